#!/usr/bin/env python3
"""Where a turn of the resident atoms_center launch goes (shader-clock ticks per stage, printed by the library): build with
    tools/build_variants.sh censt:"-DGR_EXP_STEPTIME"   and run   GR_STEPTIME=1 GR_LIB_PATH=tools/bin/ab_censt.so python tools/cen_steptime.py"""
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
    for k in range(4):
        t = time.perf_counter(); s.atoms_center_batch("all", 0, NF, weighted=True); print(bname, "call", k, round((time.perf_counter() - t) / NF * 1e6, 2), "us/frame", file=sys.stderr, flush=True)
    s.close()
