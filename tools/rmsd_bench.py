#!/usr/bin/env python3
"""calc_rmsd (RMSD without fit) on resident frames: us per frame for the f32-chain pass (k_sums_pk<false, true>) and the exact-product
pass (k_rmsd_accum<0>) over a sweep of GR_TUNE_CHUNKS; 1e6 atoms, 256 frames per call.   python tools/rmsd_bench.py [atoms] [frames]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chunk_list = [int(x) for x in os.environ.get("CHUNKS", "0 4 6 8 16 24 32 64 128 256").split()]
out = {"n_atoms": n, "frames_per_call": NF, "results": []}
for bname, box in (("dodecahedron", W.c4_box()), ("orthorhombic", W.box_from_lengths_angles([24.0, 23.0, 22.0], [90.0, 90.0, 90.0]))):
    masses = W.masses_cycle(n)
    s = G.System(n, masses=masses, n_slots=NF + 1)
    s.synth_reference(NF, box, W.blob_radius(box), 1)
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
    plan = G.RMSDPlan(ref, s, "all")
    s.sync(); time.sleep(2.0)     # (the driver clears the memory the previous section freed in the background: let that finish)
    for _ in range(6):
        plan.rmsd(0, NF)
    for fast in (1, 0):
        for ch in chunk_list:
            s.set_tuning(rmsd_fast=fast, chunks=ch)
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.4:      # (a few milliseconds of launches do not bring the device to its steady clocks)
                plan.rmsd(0, NF)
            s.profile_enable(True)
            ts = []
            for _ in range(7):
                t = time.perf_counter(); plan.rmsd(0, NF); ts.append(time.perf_counter() - t)
            prof = s.profile_read()
            out["results"].append({"box": bname, "pass": "f32 chains" if fast else "exact products", "chunks": ch,
                                   "us_per_frame_wall": round(float(np.median(ts)) / NF * 1e6, 3),
                                   "us_per_frame_kernel": round(1e3 * prof["k_sums_pk"][0] / max(prof["k_sums_pk"][2], 1), 3),
                                   "frac_hbm_12B": round(12.0 * n / (float(np.median(ts)) / NF) / 8e12, 3)})
    plan.close(); ref.close(); s.close()
print(json.dumps(out, indent=1))
