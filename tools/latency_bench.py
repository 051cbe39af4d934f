import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, groan_rs_amd as G
from groan_rs_amd import workload as W
n=32817
box=W.box_from_lengths_angles([6.44,6.76,7.26],[90.0]*3)
masses=W.masses_cycle(n)
s=G.System(n, masses=masses, n_slots=65)
s.synth_reference(64, box, 1.2, 3); s.synth_frames(64,0,64,0,0.04,3)
ref=G.System(n, masses=masses, box=box, positions=s.get_positions(64))
for x in (ref,s): x.group_create_from_ranges("Peptide",[(0,362)])
plan=G.RMSDPlan(ref,s,"Peptide")
out={}
def t(fn, reps=200):
    fn(); fn(); s.sync()
    t0=time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter()-t0)/reps*1e6
out["rmsd 1 frame/call us"]=t(lambda: plan.rmsd(0,1))
out["rmsd 16 frames/call us per frame"]=t(lambda: plan.rmsd(0,16))/16
out["rmsd 64 frames/call us per frame"]=t(lambda: plan.rmsd(0,64))/64
out["rmsd_fit 1 frame/call us"]=t(lambda: plan.rmsd_fit(0,1))
out["rmsd_fit 64 frames/call us per frame"]=t(lambda: plan.rmsd_fit(0,64))/64
out["group_get_com 1 frame us"]=t(lambda: s.group_get_com("Peptide", slot=0))
out["group_get_com_batch 64 us per frame"]=t(lambda: s.group_get_com_batch("Peptide",0,64))/64
print(json.dumps(out,indent=1))
