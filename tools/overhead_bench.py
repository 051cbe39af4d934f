#!/usr/bin/env python3
"""Fixed cost of a batched call: a 512-atom group in a 1e6-atom system, 256 frames per call -- the kernels are microseconds, what is
left is the host: checks, state upload, launches, read-back, synchronisation.   python tools/overhead_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
box = W.c4_box()
masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=NF + 1)
s.synth_reference(NF, box, W.blob_radius(box), 1)
s.synth_frames(NF, 0, NF, 0, 0.05, 1)
s.group_create_from_ranges("small", [(0, 511)])
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
ref.group_create_from_ranges("small", [(0, 511)])
plan = G.RMSDPlan(ref, s, "small")
out = {}
def timed(name, fn, reps=40):
    for _ in range(5): fn()
    s.sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    out[name] = {"us_per_call_median": round(1e6 * float(np.median(ts)), 1), "us_per_call_min": round(1e6 * float(np.min(ts)), 1)}
timed("group_get_com_batch(512 atoms, 256 frames)", lambda: s.group_get_com_batch("small", 0, NF))
timed("group_center_naive_batch", lambda: s.group_center_batch("small", G._lib.CENTER_NAIVE, 1, 0, NF))
timed("group_estimate_com_batch", lambda: s.group_estimate_com_batch("small", 0, NF))
timed("rmsd_batch", lambda: plan.rmsd(0, NF))
timed("rmsd_fit_batch (fit of 1e6 atoms: NOT overhead)", lambda: plan.rmsd_fit(0, NF), reps=5)
timed("sync only", lambda: s.sync())
print(json.dumps(out, indent=1))
