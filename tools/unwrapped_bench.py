#!/usr/bin/env python3
"""Frames that were never wrapped: whole molecules (blocks of 1000 atoms) displaced by up to +-3 box vectors, as a "nojump" trajectory
delivers them.  us per frame of gr_rmsd_fit_batch / gr_rmsd_batch / get_com / atoms_wrap against the same frames wrapped into the cell;
1e6 atoms, 64 frames per call, default tuning and the two-pass path.   python tools/unwrapped_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 64
box = W.c4_box()
masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=NF + 1)
s.synth_reference(NF, box, W.blob_radius(box), 1)
s.synth_frames(NF, 0, NF, 0, 0.05, 1)
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
plan = G.RMSDPlan(ref, s, "all")
rng = np.random.default_rng(2)
L = W.box_matrix(box)
wrapped = [s.get_positions(f) for f in range(8)]
unwrapped = []
for f in range(8):
    k = rng.integers(-3, 4, (n // 1000 + 1, 3)).repeat(1000, axis=0)[:n]
    unwrapped.append((wrapped[f].astype(np.float64) + k @ L).astype(np.float32))
out = {"n_atoms": n, "frames_per_call": NF}
def timed(fn, reps=5):
    fn(); s.sync()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return round(1e6 * float(np.median(ts)) / NF, 3)
for kind, frames in (("wrapped", wrapped), ("unwrapped", unwrapped)):
    def load():
        for f in range(NF):
            s.set_frame(frames[f % 8], box, slot=f)
    for mode, tune in (("default", dict(resident=1)), ("two-pass", dict(resident=0))):
        s.set_tuning(**tune)
        load(); t0 = time.perf_counter(); r, st = plan.rmsd_fit(0, NF); dt = time.perf_counter() - t0      # fresh frames: ONE call (a second call would fit fitted frames)
        out["%s / rmsd_fit %s us/frame (one call)" % (kind, mode)] = round(1e6 * dt / NF, 3)
        out["%s / rmsd_fit %s fallbacks" % (kind, mode)] = plan.last_fallbacks()
    load()
    out["%s / rmsd us/frame" % kind] = timed(lambda: plan.rmsd(0, NF))
    out["%s / get_com us/frame" % kind] = timed(lambda: s.group_get_com_batch("all", 0, NF))
    out["%s / estimate_com us/frame" % kind] = timed(lambda: s.group_estimate_com_batch("all", 0, NF))
    load(); t0 = time.perf_counter(); s.group_wrap_batch(None, 0, NF); out["%s / atoms_wrap us/frame (one call)" % kind] = round(1e6 * (time.perf_counter() - t0) / NF, 3)
print(json.dumps(out, indent=1))
