#!/usr/bin/env python3
"""atoms_center about SMALL reference groups (a hundredth, a tenth, a fifth of 1e6 atoms): the one-pass resident form forced (GR_TUNE_RESIDENT 2) against the
library's choice; where the thresholds of center_resident (15 % of the system in orthorhombic cells, 3 % in others) come from."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
for bname, box in (("orthorhombic", W.box_from_lengths_angles([24.0, 23.0, 22.0], [90.0, 90.0, 90.0])), ("dodecahedron", W.c4_box())):
    s = G.System(n, masses=W.masses_cycle(n), n_slots=NF + 1)
    s.synth_reference(NF, box, 0.2 * float(min(box[:3])), 1)
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    for name, frac in (("tenth", 10), ("fifth", 5), ("hundredth", 100)):
        s.group_create_from_ranges(name, [(0, n // frac - 1)])
        for res in (2, 1, 2, 1):
            s.set_tuning(resident=res)
            l0 = s.stat("center_res_launches")
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3: s.atoms_center_batch(name, 0, NF)
            ts = []
            for _ in range(7):
                t = time.perf_counter(); s.atoms_center_batch(name, 0, NF); ts.append(time.perf_counter() - t)
            print(bname, name, "resident forced" if res == 2 else "default", "taken" if s.stat("center_res_launches") > l0 else "two passes", round(float(np.median(ts)) / NF * 1e6, 3), flush=True)
    s.close()
