#!/usr/bin/env python3
"""bench.py -- frames/s of the RMSD-fit hot path on synthetic 1e6-atom frames resident in HBM.

Workload (BASELINE.json metric): 1e6 atoms in a rhombic-dodecahedral (triclinic) box
(lengths d,d,d, angles 60,60,90, d = 24.18 nm); reference = blob of radius 0.2 x (shortest box height) about
the box centre; frame f = R_f (x0 - c) + c + t_f + noise(0.05 nm), wrapped into the cell (so the blob is
broken across the periodic boundaries); masses {1.008,12.011,14.007,15.999}[i mod 4]; group = all atoms.
One "step" = gr_rmsd_fit_batch over `--frames-per-step` frames (RMSD + in-place fit of every atom).
Every frame of warmup + timed steps is a distinct HBM-resident buffer (no reuse inside a run while it fits).

  python bench.py                      # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N   # one rank per GPU

Multi-GPU: frames are independent units, sharded round-robin (global frame g -> rank g % N, exactly the
reference's thread sharding, src/system/parallel.rs:424-448); no data-path collective; one final gather of the
per-frame RMSDs over RCCL (torch.distributed, backend nccl) inside the timed region.  Scaling is weak: each rank
processes `--frames-per-step` frames per step.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel, measured with HIP events on the
library's own stream during the timed region (gr_profile_*); `cpu_baseline` times the CPU oracle's restatement
of the reference path (oracle/, kind "port") on a bounded sample of the same frames, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
SEED = 20260424


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--atoms", type=int, default=1_000_000)
    ap.add_argument("--frames-per-step", type=int, default=1024)
    ap.add_argument("--max-pool-gb", type=float, default=160.0)
    ap.add_argument("--cpu-frames-per-thread", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(host cores visible, 16): "
                    "16 is one GPU's share of the host cores on the GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-torch", action="store_true", help="import torch first even at N=1 (coexistence check)")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: everything else any library prints there (RCCL prints a version banner on
    # init) is sent to stderr by pointing fd 1 at fd 2 until the result is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    backend = os.environ.get("GROAN_DIST_BACKEND", "nccl")   # "gloo": CPU rehearsal of the multi-rank path (ranks may share a GPU)
    tdev = None
    if world > 1 or args.with_torch:
        # torch first: its bundled HIP runtime (SONAME libamdhip64.so.7) is then the one libgroan_hip.so binds to
        import torch
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)   # plain `python bench.py --with-torch` (no launcher)
        ndev = max(torch.cuda.device_count(), 1)
        local_rank = local_rank % ndev
        if world > 1 or args.with_torch:
            import torch.distributed as dist
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                tdev = torch.device("cuda", local_rank)
                dist.init_process_group(backend="nccl", device_id=tdev)
            else:
                tdev = torch.device("cpu")
                dist.init_process_group(backend=backend)

    import numpy as np
    import groan_rs_amd as G

    G._lib.load()
    n = args.atoms
    B = args.frames_per_step
    K, W = args.steps, args.warmup
    frame_bytes = ((n + 3) // 4 * 4) * 12
    want = (K + W) * B
    pool = max(B, min(want, int(args.max_pool_gb * 1e9 // frame_bytes) // B * B))
    d = 24.18
    box = np.zeros(9, np.float32)
    # rhombic dodecahedron, SimBox::from_lengths_angles([d,d,d],[60,60,90]) (simbox.rs:96-123, :300-314)
    import math
    a, b_, g_ = [np.float32(x) * np.float32(math.pi) / np.float32(180.0) for x in (60.0, 60.0, 90.0)]
    f32 = np.float32
    box[0] = d
    box[5] = f32(d) * f32(math.cos(g_)); box[1] = f32(d) * f32(math.sin(g_))
    box[7] = f32(d) * f32(math.cos(b_))
    box[8] = f32(d) * (f32(math.cos(a)) - f32(math.cos(b_)) * f32(math.cos(g_))) / f32(math.sin(g_))
    box[2] = f32(math.sqrt(f32(d) * f32(d) - box[7] * box[7] - box[8] * box[8]))
    height = float(min(box[0], box[1], box[2]))
    radius = 0.2 * height
    masses = np.array([1.008, 12.011, 14.007, 15.999], np.float32)[np.arange(n) % 4]

    dev = local_rank
    cur = G.System(n, masses=masses, n_slots=pool + 1, device=dev)      # slot `pool` holds the reference blob
    cur.synth_reference(pool, box, radius, SEED)
    ref = G.System(n, masses=masses, n_slots=1, device=dev)
    ref.set_frame(cur.get_positions(pool), box)
    plan = G.RMSDPlan(ref, cur, "all")
    # local frame j of this rank is global frame j*world + rank
    t_gen = time.time()
    cur.synth_frames(pool, 0, pool, rank, 0.05, SEED, frame_index_stride=world)
    cur.sync()
    t_gen = time.time() - t_gen

    def barrier():
        cur.sync()
        if dist is not None:
            dist.barrier()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    rmsd_all = np.zeros((K, B), np.float32)
    step_slot = lambda s: ((s * B) % pool)
    # ---- warmup (untimed); profiling already on so the first use of the profiling events is not timed
    cur.profile_enable(True)
    for s in range(W):
        r, st = plan.rmsd_fit(step_slot(s), B)
        assert (st == 0).all(), st
    fallbacks = 0
    barrier()
    cur.profile_enable(True)
    cur.timer_start()
    t0 = time.perf_counter()
    for s in range(K):
        r, st = plan.rmsd_fit(step_slot(W + s), B)
        rmsd_all[s] = r
        fallbacks += plan.last_fallbacks()
    gathered = None
    if dist is not None:
        # final gather of the per-frame RMSDs (K*B floats per rank) over RCCL, restored to global frame order
        gathered = G.gather_per_frame(rmsd_all.reshape(-1), K * B * world, dist=dist, device=tdev)
    gpu_ms = cur.timer_stop()
    barrier()
    t1 = time.perf_counter()
    prof = cur.profile_read()
    cur.profile_enable(False)
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert np.isfinite(rmsd_all).all()
    total_frames = K * B * world
    value = total_frames / elapsed

    # ---- roofline of the dominant kernel (HIP events on the library's stream, timed region)
    # algorithmic bytes per frame: the persistent kernel is the whole path (read x, p, m once + write x = 40 B/atom)
    alg_bytes = {"k_rmsd_accum": 28.0 * n, "k_rmsd_finalize": 0.0, "k_fit": 24.0 * n, "k_rmsd_fit_persist": 40.0 * n}
    dom = max(("k_rmsd_accum", "k_fit", "k_rmsd_fit_persist"), key=lambda k: prof[k][0] if k in prof else -1.0)
    ms_total, launches, frames = prof[dom]
    avg_ms = ms_total / max(launches, 1)
    bytes_per_launch = alg_bytes[dom] * (frames / max(launches, 1))
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (separate --pmc runs,
    # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; profiles/*_pmc_summary.json) -- only when the launch shape matches
    traffic, traffic_src = None, None
    try:
        import glob
        for pf in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")), reverse=True):
            pj = json.load(open(pf))
            if pj.get("n_atoms") == n and pj.get("frames_per_launch") == frames / max(launches, 1):
                key = [k for k in pj["traffic"] if k.startswith(dom)]
                if key:
                    traffic = pj["traffic"][key[0]]["hbm_bytes_per_launch"]; traffic_src = os.path.basename(pf)
                    break
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 4), "launches": launches, "frames_per_launch": frames / max(launches, 1),
                "algorithmic_bytes_per_launch": bytes_per_launch}
    kernels = {k: {"ms_total": round(v[0], 3), "launches": v[1], "us_per_frame": round(1e3 * v[0] / max(v[2], 1), 3)} for k, v in prof.items()}
    path_gbs = 40.0 * n * (K * B) / (gpu_ms * 1e-3) / 1e9   # whole step, 40 B/atom/frame algorithmic

    out = {
        "metric": "frames/sec RMSD-fit, 1e6 atoms triclinic, 1/2/4/8 GPUs; HBM GB/s vs peak",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (fp64 accumulators)", "data": "synthetic",
        "config": {"workload": "synthetic %d-atom rhombic-dodecahedral (triclinic) frames resident in HBM, Kabsch RMSD-fit of all atoms "
                               "(BASELINE configs[3] shard per GPU)" % n,
                   "n_atoms": n, "frames_per_step": B, "frames_per_gpu": K * B, "selection": "all atoms", "box9": [float(x) for x in box],
                   "pool_frames": pool, "parallelism": "frames round-robin over %d GPU(s), final RCCL gather" % world,
                   "fallback_frames": fallbacks, "synth_seconds": round(t_gen, 2)},
        "roofline": roofline,
        "kernels": kernels,
        "path": {"algorithmic_GBs": round(path_gbs, 1), "frac_of_peak": round(path_gbs / HBM_PEAK_GBS, 4), "gpu_ms_timed_region": round(gpu_ms, 3),
                 "bytes_per_frame": 40.0 * n},
    }

    # ---- CPU baseline: the oracle's restatement of the reference path on a bounded sample (rank 0, N=1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            import oracle_lib as O
            cores = args.cpu_threads or min(len(os.sched_getaffinity(0)), 16)
            nf = max(cores * args.cpu_frames_per_thread, 2)
            nf = min(nf, pool)
            # fresh frames: regenerate the first nf frames (the timed region fitted them in place)
            cur.synth_frames(pool, 0, nf, rank, 0.05, SEED, frame_index_stride=world)
            sample = np.stack([cur.get_positions(f) for f in range(nf)])
            ref_pos = ref.get_positions(0)
            gpu_r, gpu_st = plan.rmsd_fit(0, nf)
            gpu_fit0 = cur.get_positions(0)
            soa = sample.copy()
            sec_f, r_f = O.baseline_rmsd_fit(sample, ref_pos, masses, box, cores, 0)
            sec_s, r_s = O.baseline_rmsd_fit(soa, ref_pos, masses, box, cores, 1)
            sec_1, _ = O.baseline_rmsd_fit(soa[:1].copy(), ref_pos, masses, box, 1, 0)   # one thread, one (already fitted) frame
            # parity at full size: the reference's sequential f32 sums lose ~1e-2 nm over 1e6 terms, so the GPU (fp64
            # sums) is compared with the same CPU restatement summing in double; the f32 figure is reported beside it
            cur.synth_frames(pool, 0, 2, rank, 0.05, SEED, frame_index_stride=world)
            chk = np.stack([cur.get_positions(f) for f in range(2)])
            O.set_accumulate_f64(True)
            _, r_64 = O.baseline_rmsd_fit(chk, ref_pos, masses, box, 2, 1)
            O.set_accumulate_f64(False)
            out["cpu_baseline"] = {
                "value": round(nf / sec_f, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": "%d of the benchmark's own frames (%d per thread), reference-faithful 232-byte AoS layout, f32, frames round-robin over %d threads" % (nf, args.cpu_frames_per_thread, cores),
                "soa_value": round(nf / sec_s, 2), "single_thread_value": round(1.0 / sec_1, 3),
                "gpu_vs_cpu": round(value / (nf / sec_f), 1),
                "parity_max_abs_rmsd_diff_vs_cpu_f64sums": float(np.abs(gpu_r[:2] - r_64).max()),
                "parity_max_abs_fit_diff_vs_cpu_f64sums_frame0": float(np.abs(gpu_fit0 - chk[0]).max()),
                "reference_f32_sum_error_rmsd": float(np.abs(r_f[:2] - r_64).max()),
            }
        except Exception as e:   # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        if gathered is not None:
            out["config"]["gathered_frames"] = int(gathered.shape[0])
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    plan.close(); ref.close(); cur.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
