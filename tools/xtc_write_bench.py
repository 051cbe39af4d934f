#!/usr/bin/env python3
"""Fitted-trajectory output: gr_xtc_write_slots over resident frames, device encoder against host threads (16), frames per second
and MB/s of xtc written.  5e5 atoms (BASELINE configs[4]'s size) water-like and protein-like (short runs) frames, 256 per call.
    python tools/xtc_write_bench.py [n_atoms] [frames]"""
import json, os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(1)
L = (n / 100.0) ** (1.0 / 3.0)                                   # ~100 atoms per nm^3
box = np.array([L, L, L, 0, 0, 0, 0, 0, 0], np.float32)
def water():
    o = rng.uniform(0, L, ((n + 2) // 3, 3))
    return (np.repeat(o, 3, axis=0)[:n] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
def polymer():                                                   # a chain: every atom 0.15 nm from its predecessor
    st = rng.normal(0, 1, (n, 3)); st *= 0.15 / np.linalg.norm(st, axis=1)[:, None]
    return (np.cumsum(st, axis=0) % L).astype(np.float32)
out = {"n_atoms": n, "frames_per_call": NF, "results": {}}
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
for kind, gen in (("water-like", water), ("polymer-like", polymer)):
    s = G.System(n, n_slots=NF)
    base = gen()
    for f in range(NF):
        s.set_frame(base + np.float32(0.001 * f), box, slot=f)
    for device in (1, 0):
        s.set_tuning(xtc_device_encode=device)
        best = None
        for rep in range(3):
            path = os.path.join(tmp, "w.xtc")
            with G.XtcWriter(path) as w:
                s.sync(); t0 = time.perf_counter()
                w.write_slots(s, 0, NF, precision=1000.0, host_threads=16)
                dt = time.perf_counter() - t0
            size = os.path.getsize(path); os.remove(path)
            best = dt if best is None else min(best, dt)
        out["results"]["%s, %s" % (kind, "device encoder" if device else "16 host encoders")] = {"frames_per_s": round(NF / best, 1), "ms_per_frame": round(1e3 * best / NF, 3), "xtc_MB_per_s": round(size / best / 1e6, 1), "bytes_per_atom": round(size / NF / n, 3)}
    s.close()
print(json.dumps(out, indent=1))
