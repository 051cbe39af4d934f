#!/usr/bin/env python3
"""Cut-off pair search (gr_group_pairs_within, device cell grid) at 1e6 atoms uniform in an orthorhombic cell:
ms per call and pairs/s, for the whole system against itself and for a 1e4-atom group against everything."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import groan_rs_amd as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
box = np.array([24.0, 23.0, 22.0, 0, 0, 0, 0, 0, 0], np.float32)
s = G.System(n, n_slots=1)
s.synth_uniform(0, box, 20260424)
s.group_create_from_ranges("S", [(0, 9999)])
out = {}
lib = s._lib
for label, g1, g2, cutoff in (("all x all, 0.35 nm", "all", "all", 0.35), ("1e4 x all, 0.6 nm", "S", "all", 0.6), ("all x all, 0.6 nm (count only)", "all", "all", 0.6)):
    cnt = C.c_uint64(0)
    lib.gr_group_pairs_within(s._ctx, 0, g1.encode(), g2.encode(), C.c_float(cutoff), 0, None, None, None, C.byref(cnt))
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        lib.gr_group_pairs_within(s._ctx, 0, g1.encode(), g2.encode(), C.c_float(cutoff), 0, None, None, None, C.byref(cnt))
    t_count = (time.perf_counter() - t0) / reps
    entry = {"pairs": int(cnt.value), "count_only_ms": round(1e3 * t_count, 2)}
    if "count only" not in label:
        t0 = time.perf_counter()
        i, j, d = s.group_pairs_within(g1, g2, cutoff)
        entry["count_write_readback_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
        entry["Mpairs_per_s"] = round(i.size / (time.perf_counter() - t0) / 1e6, 1)
    out[label] = entry
print(json.dumps(out, indent=1))
