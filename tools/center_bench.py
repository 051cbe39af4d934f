#!/usr/bin/env python3
"""atoms_center(_mass) of the whole system, NF resident frames per call: the one-pass resident form (GR_TUNE_CENTER_RESIDENT 1) against the two passes (0);
wall us per frame of every call.   python tools/center_bench.py [atoms] [frames]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 256
out = {"n_atoms": n, "frames_per_call": NF, "results": []}
for bname, box in (("orthorhombic", W.box_from_lengths_angles([24.0, 23.0, 22.0], [90.0, 90.0, 90.0])), ("dodecahedron", W.c4_box())):
    masses = W.masses_cycle(n)
    s = G.System(n, masses=masses, n_slots=NF + 1)
    s.synth_reference(NF, box, 0.2 * float(min(box[:3])), 1)
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    s.group_create_from_ranges("half", [(0, n // 2)])
    s.group_create_from_ranges("tenth", [(0, n // 10 - 1)])
    for grp, weighted in (("all", True), ("all", False), ("half", True), ("tenth", False)):
        for mode in (1, 0, 1, 0):
            s.set_tuning(center_resident=mode)
            l0 = s.stat("center_res_launches")
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                s.atoms_center_batch(grp, 0, NF, weighted=weighted)
            ts = []
            for _ in range(7):
                t = time.perf_counter(); s.atoms_center_batch(grp, 0, NF, weighted=weighted); ts.append(time.perf_counter() - t)
            r = {"box": bname, "reference_group": grp, "weighted": weighted, "center_resident": mode, "taken": s.stat("center_res_launches") > l0,
                 "us_per_frame": round(float(np.median(ts)) / NF * 1e6, 3), "calls": [round(t / NF * 1e6, 2) for t in ts], "turn_ns": None}
            out["results"].append(r); print(json.dumps(r), file=sys.stderr, flush=True)
    s.close()
print(json.dumps(out, indent=1))
