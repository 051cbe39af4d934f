#!/usr/bin/env python3
"""Fixed cost of a batched call: wall time of a call against the number of frames in it (least-squares line: us per call = a + b frames), 1e6 atoms."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NFMAX = 1_000_000, 1024
box = W.c4_box()
s = G.System(n, masses=W.masses_cycle(n), n_slots=NFMAX + 1)
s.synth_reference(NFMAX, box, 0.2 * float(min(box[:3])), 1)
s.synth_frames(NFMAX, 0, NFMAX, 0, 0.05, 1)
s.group_create_from_ranges("tenth", [(0, n // 10 - 1)])
ref = G.System(n, masses=W.masses_cycle(n), box=box, positions=s.get_positions(NFMAX))
plan = G.RMSDPlan(ref, s, "all")
ops = {"atoms_center_mass(all)": lambda nf: s.atoms_center_batch("all", 0, nf, weighted=True),
       "get_com(tenth)": lambda nf: s.group_get_com_batch("tenth", 0, nf),
       "get_com(all)": lambda nf: s.group_get_com_batch("all", 0, nf),
       "atoms_wrap": lambda nf: s.group_wrap_batch(None, 0, nf),
       "calc_rmsd(all)": lambda nf: plan.rmsd(0, nf),
       "rmsd_fit(all)": lambda nf: plan.rmsd_fit(0, nf)}
for name, fn in ops.items():
    xs, ys = [], []
    for nf in (64, 128, 256, 512, 1024):
        for _ in range(3): fn(nf)
        ts = []
        for _ in range(9):
            t = time.perf_counter(); fn(nf); ts.append(time.perf_counter() - t)
        xs.append(nf); ys.append(float(np.median(ts)) * 1e6)
    b, a = np.polyfit(xs, ys, 1)
    print("%-24s us per call = %6.1f + %.3f x frames   (%s)" % (name, a, b, ", ".join("%d: %.0f" % (x, y) for x, y in zip(xs, ys))), flush=True)
