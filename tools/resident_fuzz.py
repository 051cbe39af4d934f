#!/usr/bin/env python3
"""Randomised soak of the resident RMSD-fit pass on the GPU box: system sizes from 3 000 to 1 000 000 atoms, 1-200 frames per call,
forced and automatic numbers of frame streams, whole, nearly-whole and 45-98 % selections, orthorhombic / triclinic / dodecahedral cells, a
box per frame now and then, a frame without a position now and then -- every call compared with the two-pass path on the same
frames (rmsd to 2e-6 nm, fitted coordinates to 2e-5 nm, statuses equal).  Prints one line per case; exit status 1 on a mismatch.

    python tools/resident_fuzz.py [seconds] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CELLS = [([7.0, 6.5, 6.0], [90.0, 90.0, 90.0]), ([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]), ([6.5, 6.5, 6.5], [60.0, 60.0, 90.0])]
t_end = time.time() + budget
case = bad = 0
stats = {"resident": 0, "two_pass": 0, "aborts": 0, "misses": 0}
while time.time() < t_end:
    case += 1
    n = int(np.exp(rng.uniform(np.log(3_000), np.log(1_000_000))))
    nf = int(rng.choice([1, 2, 5, 7, 16, 33, 64, 100, 200]))
    if n * nf > 60_000_000:
        nf = max(1, 60_000_000 // n)
    streams = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 13, 32]))
    forced = bool(rng.integers(0, 2)) or streams != 0
    l, a = CELLS[int(rng.integers(0, 3))]
    scale = (n / 20_000.0) ** (1.0 / 3.0)
    box = W.box_from_lengths_angles([x * scale for x in l], a)
    whole = bool(rng.integers(0, 2))
    if whole:
        sel = (0, n - 1)
    elif rng.integers(0, 2):
        sel = (int(rng.integers(0, max(1, n // 30))), n - 1 - int(rng.integers(0, max(1, n // 30))))      # nearly the whole system
    else:
        m = int(n * rng.uniform(0.45, 0.98))                                                              # 45 .. 98 % of it, anywhere (the default takes the pass from 45 %)
        a0 = int(rng.integers(0, n - m + 1))
        sel = (a0, a0 + m - 1)
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, 0.2 * float(min(box[:3])), 7 + case)
    cur.synth_frames(nf, 0, nf, 0, 0.04, 7 + case)
    ref = G.System(n, masses=masses, box=box, positions=cur.get_positions(nf))
    for s_ in (ref, cur):
        s_.group_create_from_ranges("S", [sel])
    per_frame_box = bool(rng.integers(0, 4) == 0)
    boxes_f = [W.box_from_lengths_angles([x * scale * (1.0 + 1e-4 * ((f * 5) % 7 - 3)) for x in l], a) for f in range(nf)] if per_frame_box else None
    if per_frame_box:
        for f in range(nf):
            cur.set_box(boxes_f[f], slot=f)
    bad_frame = int(rng.integers(0, nf)) if (nf > 2 and rng.integers(0, 4) == 0) else -1
    if bad_frame >= 0:
        p = cur.get_positions(bad_frame); p[int(rng.integers(sel[0], sel[1] + 1))] = np.nan
        cur.set_frame(p, cur.get_box(bad_frame), slot=bad_frame)
    # now and then a frame whose image proof fails (a group wider than half the cell): the launch must hand it back untouched
    wide_frame = int(rng.integers(0, nf)) if (nf > 2 and n * nf <= 6_000_000 and rng.integers(0, 3) == 0) else -1
    if wide_frame >= 0 and wide_frame != bad_frame:
        wide = W.proof_failing_frame(ref.get_positions(0), cur.get_box(wide_frame), "two_lobes" if rng.integers(0, 2) else "stretched", 11 + case)
        cur.set_frame(wide, cur.get_box(wide_frame), slot=wide_frame)
    else:
        wide_frame = -1
    keep = [cur.get_positions(f) for f in range(nf)] if n * nf <= 6_000_000 else None
    plan = G.RMSDPlan(ref, cur, "S")
    res = {}
    for mode in ("resident", "two_pass"):
        if mode == "two_pass":
            if keep is not None:
                for f in range(nf):
                    cur.set_frame(keep[f], cur.get_box(f), slot=f)
            else:
                if bad_frame >= 0:
                    break
                cur.synth_frames(nf, 0, nf, 0, 0.04, 7 + case)      # (too large to keep on the host: generated again -- and with them the boxes)
                if per_frame_box:
                    for f in range(nf):
                        cur.set_box(boxes_f[f], slot=f)
            cur.set_tuning(resident=0)
        else:
            # round 5's knobs, drawn per case: the order of a turn (by fill / fit first / sums first) and the metronome (controller / off / a
            # period the launch keeps easily / one it cannot keep): they move work in time, never in value
            order, metro = int(rng.integers(0, 3)), int(rng.choice([0, 1, 20_000, 300]))
            cur.set_tuning(resident=2 if forced else 1, resident_streams=streams, resident_fit_last=order, resident_metro_ns=metro)
        cur.profile_enable(True)
        r, st = plan.rmsd_fit(0, nf, raise_on_error=False)
        prof = cur.profile_read()
        check = [0, nf // 2, nf - 1] if keep is None else range(nf)
        res[mode] = (np.array(r), np.array(st), {f: cur.get_positions(f) for f in check}, prof["k_fit_resident"][1])
    ok = True
    why = ""
    if "two_pass" in res:
        a_, b_ = res["resident"], res["two_pass"]
        if not np.array_equal(a_[1], b_[1]): ok, why = False, "statuses %s vs %s" % (a_[1], b_[1])
        good = a_[1] == 0
        if ok and good.any() and np.abs(a_[0][good] - b_[0][good]).max() > 2e-6: ok, why = False, "rmsd %g" % np.abs(a_[0][good] - b_[0][good]).max()
        for f in a_[2]:
            if ok and not np.allclose(a_[2][f], b_[2][f], atol=2e-5, rtol=0, equal_nan=True): ok, why = False, "positions of frame %d: %g" % (f, np.nanmax(np.abs(a_[2][f] - b_[2][f])))
    ran = res["resident"][3]
    stats["resident" if ran else "two_pass"] += 1
    stats["aborts"] += cur.stat("res_aborts"); stats["misses"] += cur.stat("res_handshake_misses")
    if ok and ran and wide_frame >= 0 and plan.last_fallbacks() < 1: ok, why = False, "the wide frame %d was not handed back" % wide_frame
    print("case %3d n=%7d nf=%3d streams=%2d(%d) forced=%d whole=%d cell=%s boxes=%d bad=%d wide=%d resident=%d %s %s" % (
        case, n, nf, streams, cur.stat("res_last_streams"), forced, whole, a, per_frame_box, bad_frame, wide_frame, ran, "ok" if ok else "MISMATCH", why), flush=True)
    bad += 0 if ok else 1
    plan.close(); ref.close(); cur.close()
print("cases %d, mismatches %d, %s" % (case, bad, stats), flush=True)
sys.exit(1 if bad else 0)
