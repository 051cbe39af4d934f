#!/usr/bin/env python3
"""The trajectory-fitting converter end to end on one GPU: xtc in -> RMSD-fit of every frame to the first one -> fitted xtc out (the
reference's RMSDConverterAnalyzer + XtcWriter loop, src/system/rmsd.rs:169-260 with src/io/xtc_io/mod.rs:256-331), in batches of B
frames: gr_xtc_read_frames_device (compressed stream over PCIe, unpacked on the GPU) -> gr_rmsd_fit_batch -> gr_xtc_write_slots
(compressed on the GPU, stream back over PCIe, written).  A water-like system of --atoms atoms, --frames frames; stages timed alone and
the loop as a whole; the output is read back and checked frame by frame against the fitted coordinates (one quantum).
    python tools/convert_bench.py [--atoms 500000] [--frames 512] [--batch 128]"""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W

ap = argparse.ArgumentParser()
ap.add_argument("--atoms", type=int, default=500_000); ap.add_argument("--frames", type=int, default=512); ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--file-frames", type=int, default=64)
a = ap.parse_args()
n, NF, B, FF = a.atoms - a.atoms % 3, a.frames, a.batch, a.file_frames
L = (n / 100.0) ** (1.0 / 3.0)
box = W.box_from_lengths_angles([L, L, L], [90.0] * 3)
rng = np.random.default_rng(5)
ctr = rng.uniform(0.2 * L, 0.8 * L, (n // 3, 3))
base = np.repeat(ctr, 3, axis=0); base[1::3] += rng.normal(0, 0.055, (n // 3, 3)); base[2::3] += rng.normal(0, 0.055, (n // 3, 3))
masses = np.array([15.999, 1.008, 1.008], np.float32)[np.arange(n) % 3]
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
src, dst = os.path.join(tmp, "in.xtc"), os.path.join(tmp, "fit.xtc")
s = G.System(n, masses=masses, n_slots=max(2 * B, 8))
def rot(k):
    th = 0.01 * k; c_, s_ = np.cos(th), np.sin(th)
    return np.array([[c_, -s_, 0], [s_, c_, 0], [0, 0, 1.0]])
with G.XtcWriter(src) as w:                                     # the input: the system tumbling and jittering about the cell's centre
    for f0 in range(0, FF, 8):
        for k in range(8):
            p = (base - 0.5 * L) @ rot(f0 + k).T + 0.5 * L + rng.normal(0, 0.01, base.shape)
            s.set_frame(p.astype(np.float32), box, slot=k)
        w.write_slots(s, 0, 8, steps=np.arange(f0, f0 + 8, dtype=np.int64), times=np.arange(f0, f0 + 8, dtype=np.float32), precision=1000.0)
x = G.XtcFile(src)
ref = G.System(n, masses=masses, box=box, positions=x.read_frame(0)[0])
plan = G.RMSDPlan(ref, s, "all")
out = {"n_atoms": n, "frames": NF, "batch": B, "in_MB_per_frame": round(os.path.getsize(src) / FF / 1e6, 3)}
def batches():
    for f0 in range(0, NF, B):
        yield f0, min(B, NF - f0)
def read(f0, nb, slot0=0):
    done = 0
    while done < nb:                                            # (the file's frames, round and round)
        k0 = (f0 + done) % FF; m = min(nb - done, FF - k0)
        x.read_frames_device(s, k0, m, first_slot=slot0 + done, host_threads=16); done += m
# stages alone
read(0, B, 0); read(0, B, B); s.sync()                          # (the reader's pinned banks exist now)
t0 = time.perf_counter()
for k, (f0, nb) in enumerate(batches()): read(f0, nb, (k % 2) * B)
s.sync(); out["read_alone_frames_per_s"] = round(NF / (time.perf_counter() - t0), 1)
t0 = time.perf_counter()
for f0, nb in batches(): plan.rmsd_fit(0, nb)
out["fit_alone_frames_per_s"] = round(NF / (time.perf_counter() - t0), 1)
read(0, B)
with G.XtcWriter(dst) as w: w.write_slots(s, 0, B, precision=1000.0)          # (the writer's pinned banks exist now)
t0 = time.perf_counter()
with G.XtcWriter(dst) as w:
    for f0, nb in batches(): w.write_slots(s, 0, nb, precision=1000.0)
out["write_alone_frames_per_s"] = round(NF / (time.perf_counter() - t0), 1)
# the loop, over two sets of slots: the next batch's read is issued (skim on 16 host threads, the stream over PCIe and the unpack kernel on
# the copy stream: asynchronous from there on) before this one is fitted, compressed and written
t0 = time.perf_counter()
rm = []
bl = list(batches())
with G.XtcWriter(dst) as w:
    read(bl[0][0], bl[0][1], 0)
    for k, (f0, nb) in enumerate(bl):
        cur = (k % 2) * B
        if k + 1 < len(bl): read(bl[k + 1][0], bl[k + 1][1], ((k + 1) % 2) * B)
        r, st = plan.rmsd_fit(cur, nb); rm.extend(r)
        w.write_slots(s, cur, nb, steps=np.arange(f0, f0 + nb, dtype=np.int64), times=np.arange(f0, f0 + nb, dtype=np.float32), precision=1000.0)
wall = time.perf_counter() - t0
last_slot0 = ((len(bl) - 1) % 2) * B
out["convert_frames_per_s"] = round(NF / wall, 1)
out["slowest_stage_over_wall"] = round(out["convert_frames_per_s"] / min(out["read_alone_frames_per_s"], out["fit_alone_frames_per_s"], out["write_alone_frames_per_s"]), 3)
out["out_MB_per_frame"] = round(os.path.getsize(dst) / NF / 1e6, 3)
out["device_encoded_frames"] = s.stat("xtc_device_frames")
# check: the last batch, frame by frame, against the fitted coordinates still in the slots
y = G.XtcFile(dst)
f0, nb = list(batches())[-1]
worst = 0.0
for k in (0, nb // 2, nb - 1):
    worst = max(worst, float(np.abs(y.read_frame(f0 + k)[0] - s.get_positions(last_slot0 + k)).max()))
out["max_abs_diff_written_vs_fitted_nm"] = worst
out["rmsd_first_last"] = [float(rm[0]), float(rm[-1])]
assert y.n_frames == NF and worst <= 0.0005 + 1e-6
print(json.dumps(out, indent=1))
