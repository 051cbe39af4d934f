#!/usr/bin/env python3
"""Randomised soak of the one-pass atoms_center (gr_resident.h MODE 1) on the GPU box: 3 000 .. 1 000 000 atoms, 16-200 frames per call, forced
and automatic numbers of frame streams, orthorhombic / triclinic / dodecahedral cells (a box per frame now and then), reference groups from a
third of the system to all of it, every Dimension, mass-weighted or not, atoms far outside the cell, a frame with an atom without position now
and then -- every call compared with the two passes on the same frames: statuses equal, coordinates equal BIT FOR BIT.
    python tools/center_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CELLS = [([7.0, 6.5, 6.0], [90.0, 90.0, 90.0]), ([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]), ([6.5, 6.5, 6.5], [60.0, 60.0, 90.0])]
DIMS = [G.Dimension.X, G.Dimension.Y, G.Dimension.Z, G.Dimension.XY, G.Dimension.XZ, G.Dimension.YZ, G.Dimension.XYZ]
t_end = time.time() + budget
case = bad = taken = redone = 0
while time.time() < t_end:
    case += 1
    n = int(np.exp(rng.uniform(np.log(3_000), np.log(1_000_000))))
    nf = int(rng.choice([16, 17, 33, 64, 100, 200]))
    if n * nf > 40_000_000:
        nf = max(16, 40_000_000 // n)
    streams = int(rng.choice([0, 0, 1, 2, 3, 5, 8]))
    l, a = CELLS[int(rng.integers(0, 3))]
    scale = (n / 20_000.0) ** (1.0 / 3.0)
    box = W.box_from_lengths_angles([x * scale for x in l], a)
    masses = W.masses_cycle(n)
    s = G.System(n, masses=masses, n_slots=nf + 1)
    s.synth_reference(nf, box, float(rng.uniform(0.1, 0.45)) * float(min(box[:3])), 7 + case)
    s.synth_frames(nf, 0, nf, 0, 0.04, 7 + case)
    per_frame_box = bool(rng.integers(0, 4) == 0)
    if per_frame_box:
        for f in range(nf):
            s.set_box(W.box_from_lengths_angles([x * scale * (1.0 + 1e-4 * ((f * 5) % 7 - 3)) for x in l], a), slot=f)
    m = int(n * rng.uniform(0.31, 1.0))
    a0 = int(rng.integers(0, n - m + 1))
    if rng.integers(0, 3) == 0:
        a0, m = 0, n
    s.group_create_from_ranges("R", [(a0, a0 + m - 1)])
    dim, weighted = DIMS[int(rng.integers(0, 7))], bool(rng.integers(0, 2))
    small = n * nf <= 6_000_000
    keep = None
    if small:
        keep = [s.get_positions(f) for f in range(nf)]
        for f in range(nf):                     # a few atoms several cells away, one on the origin
            far = rng.integers(0, n, 4)
            keep[f][far] += (rng.integers(-3, 4, (4, 3)) * np.array(box[:3], np.float32)).astype(np.float32)
            keep[f][int(rng.integers(0, n))] = 0.0
        if nf > 2 and rng.integers(0, 4) == 0:
            keep[int(rng.integers(0, nf))][int(rng.integers(0, n))] = np.nan
    res = {}
    for mode in (1, 0):
        if small:
            for f in range(nf):
                s.set_frame(keep[f], s.get_box(f), slot=f)
        else:
            s.synth_frames(nf, 0, nf, 0, 0.04, 7 + case)
            if per_frame_box:
                for f in range(nf):
                    s.set_box(W.box_from_lengths_angles([x * scale * (1.0 + 1e-4 * ((f * 5) % 7 - 3)) for x in l], a), slot=f)
        s.set_tuning(center_resident=mode, resident=2 if rng.integers(0, 2) or streams else 1, resident_streams=streams, resident_fit_last=int(rng.integers(0, 3)))
        l0, r0 = s.stat("center_res_launches"), s.stat("center_res_redone")
        st = np.array(s.atoms_center_batch("R", 0, nf, dim, weighted=weighted, raise_on_error=False))
        check = range(nf) if small else [0, nf // 2, nf - 1]
        res[mode] = (st, {f: s.get_positions(f) for f in check}, s.stat("center_res_launches") - l0, s.stat("center_res_redone") - r0)
    ok, why = True, ""
    if not np.array_equal(res[1][0], res[0][0]): ok, why = False, "statuses %s vs %s" % (res[1][0], res[0][0])
    for f in res[1][1]:
        if ok and not np.array_equal(res[1][1][f], res[0][1][f], equal_nan=True): ok, why = False, "frame %d: %g" % (f, np.nanmax(np.abs(res[1][1][f] - res[0][1][f])))
    taken += res[1][2]; redone += res[1][3]
    print("case %3d n=%7d nf=%3d streams=%d cell=%s boxes=%d group=%d+%d dim=%d w=%d taken=%d redone=%d %s %s" % (case, n, nf, streams, a, per_frame_box, a0, m, int(dim), weighted, res[1][2], res[1][3], "ok" if ok else "MISMATCH", why), flush=True)
    bad += 0 if ok else 1
    s.close()
print("cases %d, mismatches %d, resident launches %d, frames handed back %d" % (case, bad, taken, redone), flush=True)
sys.exit(1 if bad else 0)
