"""One per-frame call at a time (BASELINE configs[1]'s shape: a 363-atom group in a 32 817-atom system) under `rocprofv3 --hip-trace --kernel-trace --stats`:
which HIP calls and dispatches a single gr_rmsd / gr_rmsd_fit / gr_group_center call is made of.   python tools/latency_trace.py [rmsd|fit|com] [reps]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import groan_rs_amd as G
from groan_rs_amd import workload as W
what = sys.argv[1] if len(sys.argv) > 1 else "rmsd"; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = 32817
box = W.box_from_lengths_angles([6.44, 6.76, 7.26], [90.0] * 3)
masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=3)
s.synth_reference(2, box, 1.2, 3); s.synth_frames(2, 0, 2, 0, 0.04, 3)
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(2))
for x in (ref, s): x.group_create_from_ranges("Peptide", [(0, 362)])
plan = G.RMSDPlan(ref, s, "Peptide")
fn = {"rmsd": lambda: plan.rmsd(0, 1), "fit": lambda: plan.rmsd_fit(0, 1), "com": lambda: s.group_get_com("Peptide", slot=0)}[what]
fn(); fn(); s.sync()
t0 = time.perf_counter()
for _ in range(reps): fn()
print(what, "us per call:", (time.perf_counter() - t0) / reps * 1e6, "reps", reps)
