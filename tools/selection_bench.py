#!/usr/bin/env python3
"""RMSD-fit of a contiguous selection that is not the whole system (the fit still moves every atom): us per frame of one
gr_rmsd_fit_batch over 512 fresh 1e6-atom frames, for selections of 100 %, 99.9 %, 90 %, 50 % and 10 % of the atoms, with the default
tuning, the two passes, and the resident pass forced.   python tools/selection_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 512
box = W.c4_box()
masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=NF + 1)
s.synth_reference(NF, box, W.blob_radius(box), 1)
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
out = {"n_atoms": n, "frames_per_call": NF, "results": []}
for frac, sel in (("100 %", (0, n - 1)), ("99.9 %", (500, n - 501)), ("90 %", (0, 900_000 - 1)), ("70 %", (0, 700_000 - 1)), ("50 %", (0, 500_000 - 1)), ("40 %", (0, 400_000 - 1)),
                  ("30 %", (0, 300_000 - 1)), ("20 %", (0, 200_000 - 1)), ("10 %", (0, 100_000 - 1)), ("middle 50 %", (250_000, 750_000 - 1))):
    for x in (ref, s):
        x.group_create_from_ranges("S", [sel])
    plan = G.RMSDPlan(ref, s, "S")
    s.synth_frames(NF, 0, 64, 0, 0.05, 1); plan.rmsd_fit(0, 64)          # (the plan's first call resolves its weights against the target's masses on the host)
    for mode, tune in (("two passes", 0), ("resident forced", 2), ("default", 1)):
        s.set_tuning(resident=tune)
        s.synth_frames(NF, 0, NF, 0, 0.05, 1)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:
            s.group_center_batch("all", G._lib.CENTER_NAIVE, 1, 0, 256)          # (keeps the device warm without touching the frames)
        s.profile_enable(True)
        t0 = time.perf_counter(); r, st = plan.rmsd_fit(0, NF); dt = time.perf_counter() - t0
        prof = s.profile_read()
        out["results"].append({"selection": frac, "tuning": mode, "us_per_frame": round(1e6 * dt / NF, 3), "resident_launches": prof["k_fit_resident"][1],
                               "frac_hbm_24B": round(24.0 * n / (dt / NF) / 8e12, 3)})
    plan.close()
    for x in (ref, s):
        x.group_remove("S")
print(json.dumps(out, indent=1))
