"""Call latencies at BASELINE configs[1]'s shape (a 363-atom group in a 32 817-atom system): one frame per call -- the single-wave kernels of
gr_small.h (default) against the batched kernels (GR_TUNE_SMALL_CALLS = 0) -- and batches of 16 / 64 frames.  -> JSON on stdout"""
import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, groan_rs_amd as G
from groan_rs_amd import workload as W
n = 32817
box = W.box_from_lengths_angles([6.44, 6.76, 7.26], [90.0] * 3)
masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=65)
s.synth_reference(64, box, 1.2, 3); s.synth_frames(64, 0, 64, 0, 0.04, 3)
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(64))
for x in (ref, s): x.group_create_from_ranges("Peptide", [(0, 362)])
s.group_create_from_ranges("Tail", [(400, 600)])
plan = G.RMSDPlan(ref, s, "Peptide")
out = {}
def t(fn, reps=300):
    fn(); fn(); s.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return round((time.perf_counter() - t0) / reps * 1e6, 2)
for label, small in (("single-wave kernels (default)", 4096), ("batched kernels (GR_TUNE_SMALL_CALLS = 0)", 0)):
    s.set_tuning(small_calls=small)
    o = out[label] = {}
    o["rmsd 1 frame/call us"] = t(lambda: plan.rmsd(0, 1))
    o["rmsd 16 frames/call us per frame"] = round(t(lambda: plan.rmsd(0, 16)) / 16, 3)
    o["rmsd 64 frames/call us per frame"] = round(t(lambda: plan.rmsd(0, 64)) / 64, 3)
    o["rmsd_fit 1 frame/call us"] = t(lambda: plan.rmsd_fit(0, 1))
    o["rmsd_fit 64 frames/call us per frame"] = round(t(lambda: plan.rmsd_fit(0, 64)) / 64, 3)
    o["group_get_com 1 frame us"] = t(lambda: s.group_get_com("Peptide", slot=0))
    o["group_estimate_com 1 frame us"] = t(lambda: s.group_estimate_com("Peptide", slot=0))
    o["group_get_com_naive 1 frame us"] = t(lambda: s.group_get_com_naive("Peptide", slot=0))
    o["group_get_com_batch 64 us per frame"] = round(t(lambda: s.group_get_com_batch("Peptide", 0, 64)) / 64, 3)
    o["atoms_center_mass(Peptide) 1 frame us"] = t(lambda: s.atoms_center_mass("Peptide", G.Dimension.XYZ, slot=0))
    o["group_distance 1 frame us"] = t(lambda: s.group_distance("Peptide", "Tail", G.Dimension.XYZ, slot=0))
out["small_calls"] = s.stat("small_calls"); out["small_sync_fallbacks"] = s.stat("small_sync_fallbacks")
print(json.dumps(out, indent=1))
