#!/usr/bin/env python3
"""group wrap / translate of selections that are not one block, us per 1e6-atom frame (256 frames per call): one block, every third atom,
nine atoms in ten, two blocks, everything -- with the masked-span walk (default) and on the index list (GR_TUNE_MASKED_SELECTIONS = 0).
    python tools/wrap_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
box = W.c4_box()
s = G.System(n, masses=W.masses_cycle(n), n_slots=NF + 1)
s.synth_reference(NF, box, W.blob_radius(box), 1)
s.synth_frames(NF, 0, NF, 0, 0.05, 1)
sels = {"one block of a third": [(0, 333_333)], "every third atom": [(i, i) for i in range(0, n, 3)], "nine atoms in ten": [(i, i + 8) for i in range(0, n - 10, 10)],
        "two blocks of a sixth": [(0, 166_666), (500_000, 666_666)], "two blocks of 45 %": [(0, 449_999), (500_000, 949_999)], "all": [(0, n - 1)]}
def timed(fn, reps=5):
    fn(); fn(); s.sync(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return round(1e6 * float(np.median(ts)) / NF, 3)
out = {"n_atoms": n, "frames_per_call": NF, "unit": "us per frame", "results": {}}
for name, blocks in sels.items():
    for masked in (1, 0):
        if not masked and len(blocks) == 1:
            continue
        s.set_tuning(masked_selections=masked)
        s.group_create_from_ranges("S", blocks)
        out["results"][name + ("" if masked else " -- index list")] = {"wrap": timed(lambda: s.group_wrap_batch("S", 0, NF)), "translate": timed(lambda: s.group_translate_batch("S", [0.1, 0.2, 0.3], 0, NF))}
        s.group_remove("S")
print(json.dumps(out, indent=1))
