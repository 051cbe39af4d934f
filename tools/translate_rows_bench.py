#!/usr/bin/env python3
"""translate / wrap / centring in an orthorhombic cell: one float4 per lane (GR_TUNE_TRANSLATE_ROWS 1) against the three-rows walk (0); wall us per frame,
1e6 atoms, 256 frames per call."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
box = W.c4_box() if os.environ.get("TRIC") else W.box_from_lengths_angles([24.0, 23.0, 22.0], [90.0, 90.0, 90.0])
s = G.System(n, masses=W.masses_cycle(n), n_slots=NF + 1)
s.synth_reference(NF, box, 0.2 * float(min(box[:3])), 1)
s.synth_frames(NF, 0, NF, 0, 0.05, 1)
s.group_create_from_ranges("tenth", [(0, n // 10 - 1)])
s.group_create_from_ranges("half", [(n // 4, n // 4 + n // 2)])
ops = {"atoms_translate": lambda: s.group_translate_batch(None, [0.3, -0.2, 0.1], 0, NF), "atoms_wrap": lambda: s.group_wrap_batch(None, 0, NF),
       "group_wrap(half)": lambda: s.group_wrap_batch("half", 0, NF), "atoms_center(tenth)": lambda: s.atoms_center_batch("tenth", 0, NF)}
out = []
for name, fn in ops.items():
    for rows in (1, 0, 1, 0):
        s.set_tuning(translate_rows=rows)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3: fn()
        ts = []
        for _ in range(9):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        r = {"op": name, "translate_rows": rows, "us_per_frame": round(float(np.median(ts)) / NF * 1e6, 3)}
        out.append(r); print(json.dumps(r), flush=True)
