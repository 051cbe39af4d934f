#!/usr/bin/env python3
"""one forced resident launch on a small system against the two-pass path (debugging aid): python tools/res_smoke.py [atoms] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as WL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 32
box = WL.c4_box(12.0)
masses = WL.masses_cycle(n)
cur = G.System(n, masses=masses, n_slots=2 * nf + 1, device=0)
cur.synth_reference(2 * nf, box, WL.blob_radius(box), 7)
ref = G.System(n, masses=masses, n_slots=1, device=0)
ref.set_frame(cur.get_positions(2 * nf), box)
plan = G.RMSDPlan(ref, cur, "all")
out = {}
for mode, s0 in ((0, 0), (2, nf)):
    cur.synth_frames(2 * nf, s0, nf, 0, 0.05, 7)
    cur.set_tuning(resident=mode)
    print("mode", mode, "launching", flush=True)
    r, st = plan.rmsd_fit(s0, nf)
    print("mode", mode, "status ok", bool((st == 0).all()), "rmsd", r[:3], "launches", cur.stat("res_launches"), "turn ns", cur.stat("res_last_turn_ns"), flush=True)
    out[mode] = (np.array(r), cur.get_positions(s0 + nf - 1))
print("max |d rmsd|", float(np.abs(out[0][0] - out[2][0]).max()), "max |d xyz|", float(np.abs(out[0][1] - out[2][1]).max()))
