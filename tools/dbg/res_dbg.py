import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, nf = int(sys.argv[1]) if len(sys.argv) > 1 else 70001, 6
box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [90.0, 90.0, 90.0])
masses = W.masses_cycle(n)
cur = G.System(n, masses=masses, n_slots=nf + 1)
cur.synth_reference(nf, box, 1.2, W.SEED)
cur.synth_frames(nf, 0, nf, 0, 0.04, W.SEED)
ref_pos = cur.get_positions(nf)
ref = G.System(n, masses=masses, box=box, positions=ref_pos)
for s in (ref, cur):
    s.group_create_from_ranges("S", [(0, n - 1)])
frames = [cur.get_positions(f) for f in range(nf)]
plan = G.RMSDPlan(ref, cur, "S")
out = {}
for mode in (0, 2):
    cur.set_tuning(resident=mode)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r, st = plan.rmsd_fit(0, nf, raise_on_error=False)
    out[mode] = (np.array(r), np.array(st), [cur.get_positions(f) for f in range(nf)])
    print(mode, "rmsd", np.array(r), "st", np.array(st), "fallbacks", plan.last_fallbacks())
for f in range(nf):
    d = np.abs(out[0][2][f] - out[2][2][f]).max(axis=1)
    bad = np.nonzero(d > 1e-3)[0]
    print("frame", f, "max diff", d.max(), "bad atoms", len(bad), bad[:8], bad[-4:] if len(bad) else "")
    if len(bad):
        a = bad[0]
        print("   atom", a, "two-pass", out[0][2][f][a], "resident", out[2][2][f][a], "input", frames[f][a])
def kabsch(Pm, Qm):
    pc, qc = Pm.mean(0), Qm.mean(0)
    H = (Pm - pc).T @ (Qm - qc)
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(Vt.T @ U.T))
    R = Vt.T @ np.diag([1, 1, d]) @ U.T
    return R, pc, qc
for f in range(2):
    A, B = out[0][2][f].astype(np.float64), out[2][2][f].astype(np.float64)
    R, pc, qc = kabsch(A, B)
    resid = np.sqrt((((A - pc) @ R.T + qc - B) ** 2).sum(1).mean())
    print("frame", f, "rigid residual two-pass->resident", resid, "R", np.round(R, 3).tolist(), "dc", qc - pc)
    # residual of resident output against the reference after optimal superposition
    R2, pc2, qc2 = kabsch(B, ref_pos.astype(np.float64))
    print("   resident output vs reference (superposed) rms", np.sqrt((((B - pc2) @ R2.T + qc2 - ref_pos) ** 2).sum(1).mean()))
