#!/usr/bin/env python3
"""GR_TUNE_STREAM_WGS_PER_CU on the grid-launched read-modify-write streams: translate / wrap / atoms_center of 1e6 atoms (256 frames per call) and the
two-pass RMSD-fit, us per frame at 8 (what the registers allow) .. 1 workgroups per CU.   python tools/stream_occ_sweep.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
out = {}
for bname, box in (("dodecahedron", W.c4_box()), ("orthorhombic", W.box_from_lengths_angles([24.0, 23.0, 22.0], [90.0, 90.0, 90.0]))):
    masses = W.masses_cycle(n)
    s = G.System(n, masses=masses, n_slots=NF + 1)
    s.synth_reference(NF, box, W.blob_radius(box), 1)
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    s.group_create_from_ranges("tenth", [(0, n // 10 - 1)])
    ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
    plan = G.RMSDPlan(ref, s, "all")
    ops = {"translate": lambda: s.group_translate_batch(None, [0.3, -0.2, 0.1], 0, NF), "wrap": lambda: s.group_wrap_batch(None, 0, NF),
           "atoms_center(tenth)": lambda: s.atoms_center_batch("tenth", 0, NF), "two-pass rmsd_fit(all)": lambda: plan.rmsd_fit(0, NF)}
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.6:
        s.group_wrap_batch(None, 0, NF)
    for per_cu in (8, 6, 4, 3, 2, 1):
        s.set_tuning(stream_wgs_per_cu=per_cu, resident=0)
        for name, fn in ops.items():
            fn(); fn(); s.sync()
            ts = []
            for _ in range(7):
                t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
            out.setdefault(bname, {}).setdefault(name, {})[str(per_cu)] = round(float(np.median(ts)) / NF * 1e6, 3)
    plan.close(); ref.close(); s.close()
print(json.dumps(out, indent=1))
