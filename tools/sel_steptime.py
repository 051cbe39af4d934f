import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 256
box = W.c4_box(); masses = W.masses_cycle(n)
s = G.System(n, masses=masses, n_slots=NF + 1)
s.synth_reference(NF, box, W.blob_radius(box), 1)
ref = G.System(n, masses=masses, box=box, positions=s.get_positions(NF))
for frac, sel in (("90 %", (0, 900_000 - 1)), ("middle 50 %", (250_000, 750_000 - 1))):
    for x in (ref, s): x.group_create_from_ranges("S", [sel])
    plan = G.RMSDPlan(ref, s, "S")
    s.set_tuning(resident_metro_ns=1)
    for k in range(3):
        s.synth_frames(NF, 0, NF, 0, 0.05, 1)
        t0 = time.perf_counter(); plan.rmsd_fit(0, NF); print(frac, "call", k, round((time.perf_counter() - t0) / NF * 1e6, 3), "us/frame", file=sys.stderr, flush=True)
    plan.close()
    for x in (ref, s): x.group_remove("S")
