#!/usr/bin/env python3
"""Calibration of the rounding estimate of the RMSD-without-fit pass (gr_finalize_math<.., FAST>): the pass with its guard off
(GR_TUNE_RMSD_FAST_SIGMAS = 0) against the exact-product pass on the same frames, over system sizes, noise levels (= rmsd) and cells.
Prints, per case, the largest |rmsd^2_fast - rmsd^2_exact| observed over the frames in units of the estimate sigma = 6e-8 S sqrt(20 / n).
    python tools/rmsd_calibrate.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
out = []
for n in (20_000, 60_000, 250_000, 1_000_000):
    for cell, box in (("dodecahedron", W.c4_box(24.18 * (n / 1e6) ** (1 / 3))), ("orthorhombic", W.box_from_lengths_angles([x * (n / 1e6) ** (1 / 3) for x in (24.0, 23.0, 22.0)], [90.0] * 3))):
        for noise in (0.002, 0.01, 0.05, 0.2):
            NF = 48 if n >= 250_000 else 96
            masses = W.masses_cycle(n)
            s = G.System(n, masses=masses, n_slots=NF + 1)
            s.synth_reference(NF, box, W.blob_radius(box), 5)
            s.synth_frames(NF, 0, NF, 0, noise, 5)
            ref_pos = s.get_positions(NF)
            ref = G.System(n, masses=masses, box=box, positions=ref_pos)
            plan = G.RMSDPlan(ref, s, "all")
            s.set_tuning(rmsd_fast=1, rmsd_fast_min=0, rmsd_fast_sigmas=0)
            rf, _ = plan.rmsd(0, NF)
            s.set_tuning(rmsd_fast=0)
            re, _ = plan.rmsd(0, NF)
            rf, re = np.asarray(rf, np.float64), np.asarray(re, np.float64)
            # S = (sum w|p|^2 + sum w|v|^2) / W: p about the box centre (the blob's), v about the group's first atom -- both ~ the blob's size
            p = ref_pos.astype(np.float64) - np.array([box[0], box[1], box[2]]) / 2
            v = ref_pos.astype(np.float64) - ref_pos[0].astype(np.float64)
            S = float((masses * (p * p).sum(1)).sum() / masses.sum() + (masses * (v * v).sum(1)).sum() / masses.sum())
            sigma = 6.0e-8 * S * np.sqrt(20.0 / n)
            d2 = np.abs(rf * rf - re * re)
            out.append({"n": n, "cell": cell, "noise": noise, "rmsd": round(float(re.mean()), 5), "S": round(S, 2), "sigma_r2": float(sigma),
                        "max_abs_d_r2": float(d2.max()), "rms_d_r2": float(np.sqrt((d2 * d2).mean())), "max_in_sigmas": round(float(d2.max() / sigma), 2),
                        "rms_in_sigmas": round(float(np.sqrt((d2 * d2).mean()) / sigma), 2), "max_abs_d_rmsd": float(np.abs(rf - re).max())})
            plan.close(); ref.close(); s.close()
print(json.dumps(out, indent=1))
