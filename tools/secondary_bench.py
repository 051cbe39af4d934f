#!/usr/bin/env python3
"""Secondary pipelines of SURVEY.md section 8(d) on resident frames: centres / COM, translate + wrap, centring.
1e6 atoms, NF frames per call through the batch entry points; prints one JSON object with, per operation, us per frame,
the HBM-compulsory GB/s (the frame's own bytes; arrays shared by every frame of a call count once per call) with its fraction of
the 8 TB/s HBM peak, and the algorithmic GB/s of SURVEY 8(d) (every array once per frame) without a fraction."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
import oracle_lib as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 256
REPS = 9
PEAK = 8000.0
out = {"n_atoms": n, "frames_per_call": NF}
for bname, (l, a) in {"orthorhombic": ([24.0, 23.0, 22.0], [90.0, 90.0, 90.0]), "dodecahedron": ([24.18] * 3, [60.0, 60.0, 90.0])}.items():
    box = O.box_from_lengths_angles(l, a)
    masses = np.array([1.008, 12.011, 14.007, 15.999], np.float32)[np.arange(n) % 4]
    s = G.System(n, masses=masses, n_slots=NF + 1)
    s.synth_reference(NF, box, 0.2 * float(min(box[:3])), 1)
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    s.group_create_from_ranges("tenth", [(0, n // 10 - 1)])

    calls = []

    def timed(fn):
        fn(); fn(); s.sync()
        ts = []                     # wall clock around whole calls (each ends with its own read-back + synchronisation)
        for _ in range(REPS):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        calls.append([round(t / NF * 1e6, 3) for t in ts])          # every call of the op, in order (where does a slow one sit?)
        return float(np.median(ts)) / NF * 1e6, float(np.max(ts)) / NF * 1e6     # us per frame: median, worst call

    # (callable, algorithmic bytes per frame [SURVEY 8d: every array once PER FRAME], HBM-compulsory bytes per frame [the frame's own
    #  positions read once / written once], HBM-compulsory bytes per CALL [masses, reference coordinates: the same for every frame
    #  of the call, they come from the L2 / Infinity Cache after the first frame]).  `frac_of_hbm_peak` is computed on the
    #  HBM-compulsory bytes -- it cannot exceed 1; the algorithmic rate is reported beside it without a fraction.
    ops = {
        "group_get_com_naive(all)  [16 B/atom]": (lambda: s.group_center_batch("all", G._lib.CENTER_NAIVE, 1, 0, NF), 16.0 * n, 12.0 * n, 4.0 * n),
        "group_estimate_com(all)   [16 B/atom]": (lambda: s.group_estimate_com_batch("all", 0, NF), 16.0 * n, 12.0 * n, 4.0 * n),
        "group_get_com(all)        [16 B/atom]": (lambda: s.group_get_com_batch("all", 0, NF), 16.0 * n, 12.0 * n, 4.0 * n),
        "group_get_center(all)     [12 B/atom]": (lambda: s.group_get_center_batch("all", 0, NF), 12.0 * n, 12.0 * n, 0.0),
        "group_get_com(tenth)      [16 B/atom of the group]": (lambda: s.group_get_com_batch("tenth", 0, NF), 1.6 * n, 1.2 * n, 0.4 * n),
        "atoms_translate           [24 B/atom]": (lambda: s.group_translate_batch(None, [0.3, -0.2, 0.1], 0, NF), 24.0 * n, 24.0 * n, 0.0),
        "atoms_wrap                [24 B/atom]": (lambda: s.group_wrap_batch(None, 0, NF), 24.0 * n, 24.0 * n, 0.0),
        "atoms_center(tenth)       [24 B/atom + 1.2 B/atom estimate]": (lambda: s.atoms_center_batch("tenth", 0, NF), 24.0 * n + 1.2 * n, 24.0 * n, 0.0),
        "atoms_center_mass(all)    [24 + 16 B/atom]": (lambda: s.atoms_center_batch("all", 0, NF, weighted=True), 40.0 * n, 24.0 * n, 4.0 * n),
    }
    # RMSD-fit on a sub-selection ("S = 1e5 prefix variant"): sums over S, fit over all N
    ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
    ref.group_create_from_ranges("tenth", [(0, n // 10 - 1)])
    plan_all, plan_tenth = G.RMSDPlan(ref, s, "all"), G.RMSDPlan(ref, s, "tenth")
    ops["calc_rmsd(all)            [28 B/atom]"] = (lambda: plan_all.rmsd(0, NF), 28.0 * n, 12.0 * n, 16.0 * n)
    ops["calc_rmsd_and_fit(tenth)  [24 B/atom + 28 B/atom of the group]"] = (lambda: plan_tenth.rmsd_fit(0, NF), 24.0 * n + 2.8 * n, 24.0 * n, 1.6 * n)
    ops["calc_rmsd_and_fit(all)    [40 B/atom]"] = (lambda: plan_all.rmsd_fit(0, NF), 40.0 * n, 24.0 * n, 16.0 * n)
    s.sync(); time.sleep(1.0)   # (the driver clears the gigabytes the previous section freed in the background: let that finish)
    t_w = time.perf_counter()   # ... and bring the device back to its working clocks: half a second of read-only calls (a cold device
    while time.perf_counter() - t_w < 0.5:   # measures the first operations of the list 10-20 % slow: round 4, tools/npt_bench.py)
        s.group_center_batch("all", G._lib.CENTER_NAIVE, 1, 0, NF)
    res = {}
    for name, (fn, nbytes, hbm_frame, hbm_call) in ops.items():
        fb0 = s.center_fallbacks()
        us, worst = timed(fn)
        extra = (s.center_fallbacks() - fb0) / (REPS + 2)
        gbs = nbytes / (us * 1e-6) / 1e9
        hbm = (hbm_frame + hbm_call / NF) / (us * 1e-6) / 1e9
        res[name] = {"us_per_frame": round(us, 3), "frames_per_s": round(1e6 / us, 1), "hbm_compulsory_GBps": round(hbm, 1), "frac_of_hbm_peak": round(hbm / PEAK, 3),
                     "algorithmic_GBps": round(gbs, 1), "worst_call_us_per_frame": round(worst, 3), "extra_pass_frames_per_call": round(extra, 1),
                     "calls_us_per_frame": calls[-1]}
    out[bname] = res
    plan_all.close(); plan_tenth.close(); ref.close()
    s.close()
print(json.dumps(out, indent=1))
