"""gr_ctx_set_tuning (launch geometry / path selection of the batched RMSD calls) must never change results beyond the parity
tolerance: every setting against the default on the same frames, and against the oracle.  (Round 1 read these switches from
the environment inside the library; they are explicit context state now.)"""
import numpy as np
import pytest

import oracle_lib as O
from groan_rs_amd import workload as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


@pytest.mark.parametrize("start", [0, 1003])        # selection starting on / off a 256-atom tile boundary, ragged at both ends
def test_every_tuning_gives_the_same_answers(G, start):
    n, nf = 70_001, 5
    box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [75.0, 80.0, 70.0])
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, 0.2 * 5.0, W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.04, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    sel = (start, n - 1 - (7 if start else 0))
    for s in (ref, cur):
        s.group_create_from_ranges("S", [sel])
    idx = np.arange(sel[0], sel[1] + 1)
    frames = [cur.get_positions(f) for f in range(nf)]
    with O.acc64():
        want = [O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box) for f in range(nf)]
    plan = G.RMSDPlan(ref, cur, "S")
    settings = [dict(), dict(sub_batch=2), dict(chunks=8), dict(chunks=1), dict(fit_wgs=8), dict(fuse=0), dict(two_pass=0), dict(sub_batch=1, fuse=0, fit_wgs=1)]
    for kw in settings:
        cur.set_tuning(sub_batch=256, chunks=0, fit_wgs=0, fuse=1, two_pass=1)
        cur.set_tuning(**kw)
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        r, st = plan.rmsd_fit(0, nf)
        assert (st == 0).all() and plan.last_fallbacks() == 0, kw
        for f in range(nf):
            assert abs(float(r[f]) - want[f][0]) <= 1e-5, (kw, f, float(r[f]), want[f][0])
            assert np.abs(cur.get_positions(f) - want[f][1]).max() <= 5e-5, (kw, f)
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        r2, st = plan.rmsd(0, nf)                       # rmsd without fit: the closed-form pass
        assert np.abs(r2 - r).max() <= 2e-6, kw
    with pytest.raises(G.DeviceError):
        cur.set_tuning(sub_batch=0)
    plan.close(); ref.close(); cur.close()


def test_workgroups_per_cu_of_the_streams_change_nothing(G):
    """GR_TUNE_STREAM_WGS_PER_CU keeps surplus workgroups of the grid-launched read-modify-write streams off a CU with LDS they do not use:
    translate / wrap / centring and the two-pass fit give the same bits at 8 (uncapped), 3 and 1 workgroups per CU"""
    n, nf = 50_003, 4
    box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [60.0, 60.0, 90.0])
    masses = W.masses_cycle(n)
    out = {}
    for per_cu in (0, 8, 3, 1):
        cur = G.System(n, masses=masses, n_slots=nf + 1)
        cur.synth_reference(nf, box, 1.0, W.SEED)
        cur.synth_frames(nf, 0, nf, 0, 0.04, W.SEED)
        ref = G.System(n, masses=masses, box=box, positions=cur.get_positions(nf))
        for s in (ref, cur):
            s.group_create_from_ranges("S", [(5, n - 9)])
        cur.set_tuning(stream_wgs_per_cu=per_cu, resident=0)
        plan = G.RMSDPlan(ref, cur, "S")
        r, st = plan.rmsd_fit(0, nf)
        assert (st == 0).all()
        cur.group_translate_batch(None, [0.31, -0.2, 0.15], 0, nf)
        cur.atoms_center_batch("S", 0, nf, weighted=True)
        out[per_cu] = (np.array(r).view(np.uint32), np.stack([cur.get_positions(f) for f in range(nf)]).view(np.uint32))
        plan.close(); ref.close(); cur.close()
    for per_cu in (8, 3, 1):
        assert np.array_equal(out[per_cu][0], out[0][0]) and np.array_equal(out[per_cu][1], out[0][1]), per_cu
    with pytest.raises(Exception):
        G.System(100).set_tuning(stream_wgs_per_cu=9)
