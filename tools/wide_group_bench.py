#!/usr/bin/env python3
"""A group that spans the cell (a membrane-like slab: the whole box in x and y, 4 nm thick in z): the one-pass image proof fails in
EVERY frame, so every frame of an RMSD / RMSD-fit call goes to the literal multi-pass path.  us per frame of gr_rmsd_fit_batch,
gr_rmsd_batch and gr_group_center_batch(get_com) over 64 frames.   [GR_LIB_PATH=...] python tools/wide_group_bench.py [atoms]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
NF = 64
box = W.box_from_lengths_angles([16.0, 16.0, 12.0], [90.0, 90.0, 90.0])
rng = np.random.default_rng(3)
base = np.c_[rng.random(n) * 16.0, rng.random(n) * 16.0, 4.0 + rng.random(n) * 4.0]
masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=NF + 1)
for f in range(NF):
    s.set_frame(W.wrap_into_cell(base + rng.normal(0, 0.05, base.shape) + rng.uniform(0, 16, 3) * [1, 1, 0], box), box, slot=f)
ref = G.System(n, masses=masses, box=box, positions=base.astype(np.float32))
plan = G.RMSDPlan(ref, s, "all")
out = {"n_atoms": n, "frames_per_call": NF, "library": os.environ.get("GR_LIB_PATH", "in-tree")}
def timed(name, fn, reps=5):
    fn(); s.sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    out[name] = round(1e6 * float(np.median(ts)) / NF, 2)
timed("rmsd_batch us/frame", lambda: plan.rmsd(0, NF))
out["fallbacks of the rmsd call"] = plan.last_fallbacks()
timed("rmsd_fit_batch us/frame", lambda: plan.rmsd_fit(0, NF))
timed("group_get_com_batch us/frame", lambda: s.group_get_com_batch("all", 0, NF))
print(json.dumps(out))
