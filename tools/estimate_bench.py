#!/usr/bin/env python3
"""group_estimate_com(all) at 1e6 atoms, 256 frames per call, frames wrapped into the cell; optional sweep of GR_TUNE_STREAM_WGS_PER_CU (experiment builds)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
for bname, box in (("orthorhombic", W.box_from_lengths_angles([24.0, 23.0, 22.0], [90.0, 90.0, 90.0])), ("dodecahedron", W.c4_box())):
    s = G.System(n, masses=W.masses_cycle(n), n_slots=NF + 1)
    s.synth_reference(NF, box, 0.2 * float(min(box[:3])), 1)
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    s.group_wrap_batch(None, 0, NF)
    for per_cu in [int(x) for x in os.environ.get("PER_CU", "0").split()]:
        s.set_tuning(stream_wgs_per_cu=per_cu)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3: s.group_estimate_com_batch("all", 0, NF)
        ts = []
        for _ in range(9):
            t = time.perf_counter(); s.group_estimate_com_batch("all", 0, NF); ts.append(time.perf_counter() - t)
        print(bname, "workgroups per CU", per_cu, "estimate_com", round(float(np.median(ts)) / NF * 1e6, 3), "us/frame", flush=True)
    s.close()
