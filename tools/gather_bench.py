#!/usr/bin/env python3
"""Selections that are not one block: every third atom (water oxygens), two large blocks, a thousand small blocks -- the gather paths.
us per frame (256 frames per call, 1e6 atoms) of calc_rmsd, calc_rmsd_and_fit, get_com, estimate_com, next to the same number of atoms
as ONE block, and with the masked-span passes switched off (the index-list paths of rounds 1-3).   python tools/gather_bench.py"""
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
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
sels = {
    "one block of 333 334 atoms": [(0, 333_333)],
    "every third atom (333 334 atoms)": [(i, i) for i in range(0, n, 3)],
    "two blocks of 166 667 atoms": [(0, 166_666), (500_000, 666_666)],
    "1 000 blocks of 333 atoms": [(i * 1000, i * 1000 + 332) for i in range(1000)],
}
out = {"n_atoms": n, "frames_per_call": NF, "results": {}}
def timed(fn, reps=5):
    fn(); fn(); s.sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return round(1e6 * float(np.median(ts)) / NF, 3)
for name, blocks, masked in [(k, v, m) for k, v in sels.items() for m in (1, 0)]:
    if not masked and len(blocks) == 1:
        continue
    s.set_tuning(masked_selections=masked); ref.set_tuning(masked_selections=masked)   # (read when a group is created)
    for x in (ref, s):
        x.group_create_from_ranges("S", blocks)
    plan = G.RMSDPlan(ref, s, "S")
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.3:
        s.group_center_batch("all", G._lib.CENTER_NAIVE, 1, 0, 256)
    r = {"calc_rmsd": timed(lambda: plan.rmsd(0, NF)), "get_com": timed(lambda: s.group_get_com_batch("S", 0, NF)),
         "estimate_com": timed(lambda: s.group_estimate_com_batch("S", 0, NF))}
    fits = []
    for _ in range(3):                                   # (the first call of a kind pays its buffers; every call gets fresh frames)
        s.synth_frames(NF, 0, NF, 0, 0.05, 1); s.sync()
        t0 = time.perf_counter(); plan.rmsd_fit(0, NF); fits.append(time.perf_counter() - t0)
    r["calc_rmsd_and_fit (one call, fresh frames)"] = round(1e6 * min(fits[1:]) / NF, 3)
    r["... resident launches"] = s.stat("res_launches")
    out["results"][name + ("" if masked or len(blocks) == 1 else " -- masked_selections=0: the index-list paths")] = r
    plan.close()
    for x in (ref, s):
        x.group_remove("S")
print(json.dumps(out, indent=1))
