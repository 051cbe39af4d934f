"""Frames that fill between a half and two thirds of the chip (510-700 k atoms on MI355X): two of them do not fit side by side, and
cut into 1024-group workgroups one of them leaves a third of the CUs idle.  By DEFAULT such a frame is cut into workgroups of 768
groups (three group-units per SIMD instead of four; gr_api.hip resident_wgs) and takes the resident pass; checked against the
oracle at 550 k, 600 k and 650 k atoms."""
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


@pytest.mark.parametrize("n", [550_000, 600_000, 650_000])
def test_one_stream_of_768_group_workgroups_by_default(G, n):
    nf = 24
    box = W.c4_box()
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    if cur.stat("res_max_wgs") < 250:
        pytest.skip("a device with fewer than 250 resident workgroups: the window lies elsewhere")
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    plan = G.RMSDPlan(ref, cur, "all")
    check = [0, 5, 6, 13, nf - 1]
    before = {f: cur.get_positions(f) for f in check}
    cur.profile_enable(True)
    r, st = plan.rmsd_fit(0, nf)
    prof = cur.profile_read()
    assert (st == 0).all() and plan.last_fallbacks() == 0
    assert prof["k_fit_resident"][1] == 1 and prof["k_fit_pk"][1] == 0 and cur.stat("res_last_streams") == 1, prof
    idx = np.arange(n)
    with O.acc64():
        for f in check:
            ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, before[f], masses, idx, box)
            assert abs(float(r[f]) - ro) <= 1e-5, (f, float(r[f]), ro)
            assert np.abs(cur.get_positions(f) - want).max() <= 5e-5, f
    # the same frames through workgroups of 1024 groups and through the two passes: the same results to rounding
    for tune in (dict(resident=2, resident_wg_groups=1024), dict(resident=0)):
        cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
        cur.set_tuning(**tune)
        r2, st2 = plan.rmsd_fit(0, nf)
        assert (st2 == 0).all() and np.abs(np.asarray(r2) - np.asarray(r)).max() <= 2e-6
        assert np.abs(cur.get_positions(nf - 1) - want).max() <= 5e-5
    plan.close(); ref.close(); cur.close()
