#!/usr/bin/env python3
"""Randomised soak of k_translate_wrap_rows (orthorhombic cells, one float4 per lane) against the three-rows walk: random system sizes, contiguous
selections with ragged ends, shifts of many cells, atoms on faces / far away / without position, batched and one-frame calls, centring about a
small group -- coordinates equal BIT FOR BIT, statuses equal.   python tools/translate_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
case = bad = 0
while time.time() < t_end:
    case += 1
    n = int(np.exp(rng.uniform(np.log(5), np.log(300_000))))
    nf = int(rng.integers(1, 6))
    L = rng.uniform(2.0, 30.0, 3).astype(np.float32)
    box = np.array([L[0], L[1], L[2], 0, 0, 0, 0, 0, 0], np.float32)
    s = G.System(n, masses=np.ones(n, np.float32), n_slots=nf)
    frames = []
    for f in range(nf):
        p = (rng.uniform(-0.5, 1.5, (n, 3)) * L).astype(np.float32)
        k = max(1, n // 50)
        far = rng.integers(0, n, k); p[far] += (rng.integers(-30, 31, (k, 3)) * L).astype(np.float32)
        fa = rng.integers(0, n, k); p[fa, rng.integers(0, 3, k)] = 0.0
        fa = rng.integers(0, n, k); ax = rng.integers(0, 3, k); p[fa, ax] = L[ax]
        if rng.integers(0, 3) == 0: p[rng.integers(0, n, 2), 0] = np.nan
        frames.append(p)
    a0 = int(rng.integers(0, n)); a1 = int(rng.integers(a0, n))
    s.group_create_from_ranges("S", [(a0, a1)])
    s.group_create_from_ranges("C", [(0, min(n - 1, int(rng.integers(0, 200))))])
    v = (rng.uniform(-3, 3, 3) * L * rng.choice([0.0, 1.0, 10.0])).astype(np.float32)
    op = int(rng.integers(0, 5))
    dim = G.Dimension(int(rng.integers(1, 8)))
    res = {}
    for rows in (1, 0):
        s.set_tuning(translate_rows=rows, center_resident=0)
        for f in range(nf): s.set_frame(frames[f], box, slot=f)
        if op == 0: st = s.group_translate_batch("S", v, 0, nf, raise_on_error=False)
        elif op == 1: st = s.group_wrap_batch("S", 0, nf, raise_on_error=False)
        elif op == 2: st = s.atoms_center_batch("C", 0, nf, dim, raise_on_error=False)
        else:
            st = []
            for f in range(nf):
                try:
                    (s.group_translate("S", v, slot=f) if op == 3 else s.group_wrap("S", slot=f)); st.append(0)
                except G.GroanError: st.append(1)
        res[rows] = (np.array(st), [s.get_positions(f) for f in range(nf)])
    ok = np.array_equal(res[1][0], res[0][0]) and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[1][1], res[0][1]))
    if not ok:
        bad += 1
        print("case %d n=%d nf=%d sel=%d..%d op=%d MISMATCH" % (case, n, nf, a0, a1, op), flush=True)
    s.close()
print("cases %d, mismatches %d" % (case, bad), flush=True)
sys.exit(1 if bad else 0)
