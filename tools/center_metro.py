#!/usr/bin/env python3
"""The resident atoms_center under fixed metronome periods and both step orders (GR_TUNE_RESIDENT_METRO_NS, GR_TUNE_RESIDENT_FIT_LAST): wall us per frame,
256 frames of 1e6 atoms per call.  Measured (round 5): free-running 4.54 either order; no period beats it (4.1 us: 4.57-4.59, 4.0: 4.55-4.68, 3.9: 4.78)."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
box = W.c4_box()
s = G.System(n, masses=W.masses_cycle(n), n_slots=NF + 1)
s.synth_reference(NF, box, 0.2 * float(min(box[:3])), 1)
s.synth_frames(NF, 0, NF, 0, 0.05, 1)
for T in (1, 4600, 4400, 4300, 4200, 4100, 4000, 3900, 3800, 1):
    s.set_tuning(resident_metro_ns=T)
    for order in (0, 1):
        s.set_tuning(resident_fit_last=1 + order)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3: s.atoms_center_batch("all", 0, NF, weighted=True)
        ts = []
        for _ in range(7):
            t = time.perf_counter(); s.atoms_center_batch("all", 0, NF, weighted=True); ts.append(time.perf_counter() - t)
        print("metronome", T, "ns", "sums first" if order else "fit first", round(float(np.median(ts)) / NF * 1e6, 3), "us/frame", flush=True)
