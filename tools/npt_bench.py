#!/usr/bin/env python3
"""Constant-pressure trajectories: every frame has its own (slightly breathing) box, so the resident pass runs its UBOX = false variant
(box constants as scalar loads per frame, in both stages) and the image table is built per frame on the host.  us per frame of one
gr_rmsd_fit_batch over 768 fresh 1e6-atom frames, the same box in every frame against a box per frame; + the host cost of setting
768 boxes.   python tools/npt_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 768
box = W.c4_box()
masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=NF + 1)
s.synth_reference(NF, box, W.blob_radius(box), 1)
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
plan = G.RMSDPlan(ref, s, "all")
out = {"n_atoms": n, "frames_per_call": NF}
for kind in ("same box", "box per frame", "same box", "box per frame"):
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    t_box = 0.0
    if kind == "box per frame":
        boxes = [W.c4_box(24.18 * (1.0 + 2.0e-4 * ((f * 7) % 11 - 5))) for f in range(NF)]
        t0 = time.perf_counter()
        for f in range(NF):
            s.set_box(boxes[f], slot=f)
        t_box = time.perf_counter() - t0
    s.sync()
    # (setting 768 boxes keeps the host busy for 11 ms and the device idle: warm it up again on a scratch copy of the plan's work --
    # the first version of this script timed a cold launch and read 4.7-4.9 us for a box per frame against 4.4; `bench.py
    # --box-per-frame`, which warms up for 0.5 s, shows 4.358 against 4.357)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.4:
        s.group_center_batch("all", G._lib.CENTER_NAIVE, 1, 0, 256)
    s.profile_enable(True)
    t0 = time.perf_counter(); r, st = plan.rmsd_fit(0, NF); dt = time.perf_counter() - t0
    prof = s.profile_read()
    out.setdefault(kind, []).append({"call_us_per_frame": round(1e6 * dt / NF, 3), "kernel_us_per_frame": round(1e3 * prof["k_fit_resident"][0] / max(prof["k_fit_resident"][2], 1), 3),
                                     "set_box_us_per_frame": round(1e6 * t_box / NF, 2), "ok": bool((st == 0).all())})
print(json.dumps(out, indent=1))
