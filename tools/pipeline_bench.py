#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) pipeline: host frames -> pinned staging -> H2D on the copy stream -> RMSD-fit batches.

This is the double buffer of the north star with the decoder replaced by a stand-in (a host memcpy of an already
decoded frame into the pinned staging buffer; the real xtc decoder is NEXT-1).  Two banks of B slots: while the
kernels of bank k%2 run (gr_rmsd_batch_begin), the host stages and uploads bank (k+1)%2 (gr_frame_upload on the copy
stream); per-slot events order copy-after-compute and compute-after-copy.  Reports frames/s and the stage times
measured alone, so overlap efficiency = max(stage) / wall.  NEVER the bench.py `value` (that is HBM-resident).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--atoms", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--pool", type=int, default=8)
    ap.add_argument("--decode", choices=["none", "memcpy"], default="memcpy")
    a = ap.parse_args()
    n, B = a.atoms, a.batch
    import oracle_lib as O
    box = O.box_from_lengths_angles([24.18, 24.18, 24.18], [60.0, 60.0, 90.0])
    masses = np.array([1.008, 12.011, 14.007, 15.999], np.float32)[np.arange(n) % 4]
    cur = G.System(n, masses=masses, n_slots=2 * B + 1)
    cur.synth_reference(2 * B, box, 0.2 * float(min(box[:3])), 20260424)
    ref = G.System(n, masses=masses, box=box, positions=cur.get_positions(2 * B))
    plan = G.RMSDPlan(ref, cur, "all")
    cur.synth_frames(2 * B, 0, a.pool, 0, 0.05, 20260424)
    pool = [cur.get_positions(f) for f in range(a.pool)]
    staging = [G.pinned_array((n, 3)) for _ in range(2 * B)]
    if a.decode == "none":
        for i, (arr, _) in enumerate(staging):
            np.copyto(arr, pool[i % a.pool])

    def run(do_decode, do_upload, do_compute, nbatches):
        t0 = time.perf_counter()
        inflight = False
        rm = []
        for k in range(nbatches):
            bank = k % 2
            for i in range(B):
                slot = bank * B + i
                arr = staging[slot][0]
                if do_upload:
                    cur.upload_wait(slot)
                if do_decode:
                    np.copyto(arr, pool[(k * B + i) % a.pool])
                if do_upload:
                    cur.upload_async(arr, box, slot)
            if do_compute:
                if inflight:
                    rm.append(plan.end()[0])
                plan.begin(bank * B, B, True)
                inflight = True
        if inflight:
            rm.append(plan.end()[0])
        cur.sync()
        return time.perf_counter() - t0, rm

    run(True, True, True, 4)                                   # warm-up
    t_all, rm = run(a.decode == "memcpy", True, True, a.batches)
    t_dec, _ = run(True, False, False, a.batches) if a.decode == "memcpy" else (0.0, None)
    t_h2d, _ = run(False, True, False, a.batches)
    # compute alone: frames already resident (re-fit the same banks)
    t_cmp, _ = run(False, False, True, a.batches)
    nfr = a.batches * B
    out = {"frames": nfr, "n_atoms": n, "batch": B, "decode": a.decode,
           "end_to_end_frames_per_s": round(nfr / t_all, 1), "wall_s": round(t_all, 4),
           "decode_alone_s": round(t_dec, 4), "h2d_alone_s": round(t_h2d, 4), "compute_alone_s": round(t_cmp, 4),
           "h2d_GBps": round(nfr * n * 12 / t_h2d / 1e9, 2),
           "overlap_efficiency": round(max(t_dec, t_h2d, t_cmp) / t_all, 3),
           "rmsd_mean": float(np.mean(np.concatenate(rm)))}
    print(json.dumps(out))
    for _, ptr in staging:
        G.pinned_free(ptr)


if __name__ == "__main__":
    main()
