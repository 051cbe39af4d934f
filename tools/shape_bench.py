#!/usr/bin/env python3
"""Geometry selection at 1e6 atoms: per call = mask kernel (12 B/atom read, 1 bit/atom written) + 125 KB D2H + host
block building.  Prints ms per call for a sphere and for three shapes, contiguous and scattered source groups."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
box = np.array([24.0, 23.0, 22.0, 0, 0, 0, 0, 0, 0], np.float32)
s = G.System(n, n_slots=1)
s.synth_uniform(0, box, 20260424)
s.group_create_from_indices("scattered", np.arange(0, n, 3))
sphere = G.Sphere([12.0, 11.0, 10.0], 6.0)
cyl = G.Cylinder([12.0, 2.0, 10.0], 8.0, 15.0, G.Dimension.Y)
rect = G.Rectangular([3.0, 3.0, 3.0], 18.0, 18.0, 12.0)
out = {}
for label, src, shapes in (("all/sphere", "all", [sphere]), ("all/3 shapes", "all", [sphere, cyl, rect]), ("scattered/sphere", "scattered", [sphere])):
    s.group_create_from_geometries("picked", src, shapes)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        s.group_create_from_geometries("picked", src, shapes)
    dt = (time.perf_counter() - t0) / reps
    out[label] = {"ms_per_call": round(1e3 * dt, 3), "selected": s.group_get_n_atoms("picked"), "Matoms_per_s": round(s.group_get_n_atoms(src) / dt / 1e6, 1)}
print(json.dumps(out, indent=1))
