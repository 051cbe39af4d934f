#!/usr/bin/env python3
"""debugging aid: one dumped small_fuzz case (GR_FUZZ_DUMP) through every RMSD path, rotations against an fp64 numpy Kabsch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import groan_rs_amd as G
import oracle_lib as O
d = np.load(sys.argv[1])
frames, masses, ia, box, f = d["frames"], d["masses"], d["ia"], d["box"], int(d["f"])
n, nf = frames.shape[1], frames.shape[0] - 1
masses = np.nan_to_num(masses, nan=1.0); refpos = np.nan_to_num(frames[nf], nan=1.0)
with O.acc64():
    ro, want = O.calc_rmsd_and_fit(refpos, masses, ia, box, frames[f], masses, ia, box)
def kabsch(P, Q):
    H = P.T @ Q; U, S, Vt = np.linalg.svd(H); D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))]); return U @ D @ Vt, S
# rotation the oracle applied: least squares from the group's atoms before / after
print("oracle rmsd %.7f" % ro)
def run(tag, **tune):
    s = G.System(n, masses=masses, n_slots=nf + 1)
    for k in range(nf + 1): s.set_frame(frames[k], box, slot=k)
    ref = G.System(n, masses=masses, box=box, positions=refpos)
    for x in (s, ref):
        x.group_create_from_indices("a", ia) if bool(d["scattered"]) else x.group_create_from_ranges("a", [(int(ia[0]), int(ia[-1]))])
    s.set_tuning(**tune)
    plan = G.RMSDPlan(ref, s, "a")
    r, st, R = plan.rmsd(f, 1, return_rotation=True)
    r2, st2 = plan.rmsd_fit(f, 1)
    got = s.get_positions(f)
    print("%-34s rmsd %.7f (%+.1e) fit rmsd %+.1e | fitted vs oracle %.3g (group %.3g) | R[0] %s" % (tag, r[0], r[0] - ro, r2[0] - ro, np.nanmax(np.abs(got - want)), np.nanmax(np.abs(got[ia] - want[ia])), np.array2string(R[0][0], precision=7)))
    plan.close(); s.close(); ref.close()
    return R[0], got
Rs, gs = run("single wave (small_calls=4096)", small_calls=4096)
Rb, gb = run("batched default", small_calls=0)
R1, _ = run("batched two_pass=0", small_calls=0, two_pass=0)
R2, _ = run("batched fuse=0", small_calls=0, fuse=0)
R3, _ = run("batched rmsd_fast=0", small_calls=0, rmsd_fast=0)
print("|Rs - Rb|_F %.3g  |Rs - R(two_pass=0)| %.3g  |Rs - R(fuse=0)| %.3g" % (np.linalg.norm(Rs - Rb), np.linalg.norm(Rs - R1), np.linalg.norm(Rs - R2)))
