#!/usr/bin/env python3
"""Randomised soak of the device xtc encoder against the host encoder (the files must be equal byte for byte): system sizes 10 ..
300 000 atoms, 1 .. 40 frames, mixtures of molecules (2 .. 12 atoms within a random radius), gas, chains with random step lengths,
lattices, repeated atoms, jumps of random size, coordinates scaled over five decades, precisions 1 .. 1e5, atoms without position,
group writers (blocks and lists).  Prints one line per case; exit status 1 on a mismatch.
    python tools/xtc_enc_fuzz.py [seconds] [seed]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)


def piece(m, scale):
    kind = int(rng.integers(0, 6))
    if kind == 0:                                            # molecules
        size = int(rng.integers(2, 13)); rad = scale * 10.0 ** rng.uniform(-4, -0.5)
        o = rng.uniform(0, scale, ((m + size - 1) // size, 3))
        return np.repeat(o, size, axis=0)[:m] + rng.normal(0, rad, (m, 3))
    if kind == 1: return rng.uniform(-scale, scale, (m, 3))                                              # gas
    if kind == 2: return rng.uniform(0, scale, (1, 3)) + np.cumsum(rng.normal(0, scale * 10.0 ** rng.uniform(-5, -0.5), (m, 3)), axis=0)   # chain
    if kind == 3: return np.stack(np.unravel_index(np.arange(m), (int(m ** (1 / 3)) + 2,) * 3), axis=1) * (scale / 50.0)                   # lattice
    if kind == 4: return np.repeat(rng.uniform(0, scale, (1, 3)), m, axis=0)                                                               # one point
    return rng.uniform(0, scale, (1, 3)) + np.cumsum(rng.choice([-1.0, 0.0, 1.0], (m, 3)) * scale * 10.0 ** rng.uniform(-4, -1), axis=0)   # lattice walk


t_end = time.time() + budget
case = bad = 0
while time.time() < t_end:
    case += 1
    n = int(np.exp(rng.uniform(np.log(10), np.log(300_000))))
    nf = int(max(1, min(40, rng.integers(1, 41), 6_000_000 // n)))
    nf = max(nf, 200_000 // n + 1)
    scale = float(10.0 ** rng.uniform(-2, 3))
    precision = float(rng.choice([1.0, 10.0, 100.0, 1000.0, 4321.0, 1.0e5]))
    base = []
    for f in range(min(nf, 4)):
        parts, left = [], n
        while left > 0:
            m = int(min(left, max(1, int(np.exp(rng.uniform(0, np.log(max(2, n))))))))
            parts.append(piece(m, scale)); left -= m
        p = np.concatenate(parts)[:n].astype(np.float32)
        if rng.integers(0, 3) == 0: p[rng.integers(0, n, max(1, n // 500))] = np.nan
        base.append(p)
    frames = [base[f % len(base)] for f in range(nf)]
    group = None
    if n > 100 and rng.integers(0, 3) == 0:
        if rng.integers(0, 2): group = [(int(n * 0.1), int(n * 0.8))]
        else: group = [(i, i + int(rng.integers(0, 3))) for i in range(int(rng.integers(0, 5)), n - 8, int(rng.integers(4, 9)))]
        if sum(b - a + 1 for a, b in group) * nf < 200_000: group = None
    box = np.array([9, 9, 9, 0, 0, 0, 0, 0, 0], np.float32)
    out = {}
    for device in (1, 0):
        s = G.System(n, n_slots=nf)
        s.set_tuning(xtc_device_encode=device)
        for f in range(nf): s.set_frame(frames[f], box, slot=f)
        if group: s.group_create_from_ranges("S", group)
        path = os.path.join(tmp, "f%d.xtc" % device)
        err = None
        with G.XtcWriter(path) as w:
            try: w.write_slots(s, 0, nf, group="S" if group else None, precision=precision, host_threads=8)
            except G.XtcError as e: err = e.status
        out[device] = (open(path, "rb").read(), err, s.stat("xtc_device_frames"))
        s.close(); os.remove(path)
    ok = out[1][0] == out[0][0] and out[1][1] == out[0][1] and (out[1][2] > 0 or out[1][1] is not None)
    print("case %4d n=%6d nf=%2d scale=%8.3g precision=%7g group=%s bytes=%9d err=%s device_frames=%d %s" % (case, n, nf, scale, precision, "yes" if group else "no", len(out[0][0]), out[0][1], out[1][2], "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("cases %d, mismatches %d" % (case, bad), flush=True)
sys.exit(1 if bad else 0)
