"""Dense scattered selections (every third atom, a few large blocks, a thousand small blocks: at least an eighth of the atoms between
the first and the last one) carry a bit mask beside their index list, and calc_rmsd / calc_rmsd_and_fit / get_com / get_center read
their SPAN coalesced (k_sums_pk<.., MASK>) instead of gathering atom by atom.  Against the oracle, against the gather paths
(GR_TUNE_MASKED_SELECTIONS = 0) on the same frames; a NaN in an UNSELECTED atom of the span must not reach anything, a NaN in a
selected one must be named; frames whose image proof fails; a rigid copy (handed to the exact pass); sparse selections stay on
their lists."""
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


SHAPES = {
    "every third atom": lambda n: [(i, i) for i in range(1, n, 3)],
    "two blocks": lambda n: [(5, n // 3), (n // 2 + 3, n - 7)],
    "many small blocks": lambda n: [(i * 50 + 3, i * 50 + 20) for i in range(n // 50)],
}


def _idx(blocks):
    return np.concatenate([np.arange(a, b + 1) for a, b in blocks])


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("cell", ["ortho", "dodeca"])
def test_masked_paths_match_the_oracle_and_the_gather_paths(G, shape, cell):
    n, nf = 60_000, 8
    box = W.box_from_lengths_angles([9.0, 8.0, 7.0], [90.0] * 3) if cell == "ortho" else W.c4_box(9.0)
    masses = W.masses_cycle(n)
    blocks = SHAPES[shape](n)
    idx = _idx(blocks)
    res = {}
    for masked in (1, 0):
        cur = G.System(n, masses=masses, n_slots=nf + 1)
        cur.set_tuning(masked_selections=masked, rmsd_fast_min=0)
        cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
        cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
        ref_pos = cur.get_positions(nf)
        frames = [cur.get_positions(f) for f in range(nf)]
        frames[2] = W.proof_failing_frame(ref_pos, box, "two_lobes", 9)                      # its image proof fails
        gap = np.setdiff1d(np.arange(idx[0], idx[-1] + 1), idx)                               # atoms inside the span that are NOT selected
        frames[4] = frames[4].copy(); frames[4][gap[len(gap) // 2]] = np.nan                 # a NaN there must not reach anything
        frames[5] = frames[5].copy(); frames[5][idx[len(idx) // 2]] = np.nan                 # NaN in a selected atom
        ref = G.System(n, masses=masses, box=box, positions=ref_pos)
        ref.set_tuning(masked_selections=masked)
        for s_ in (ref, cur):
            s_.group_create_from_ranges("S", blocks)
        plan = G.RMSDPlan(ref, cur, "S")
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        r, st = plan.rmsd(0, nf, raise_on_error=False)
        com, cst = cur.group_get_com_batch("S", 0, nf, raise_on_error=False)
        cen, _ = cur.group_get_center_batch("S", 0, nf, raise_on_error=False)
        rf, stf = plan.rmsd_fit(0, nf, raise_on_error=False)
        fitted = [cur.get_positions(f) for f in range(nf)]
        res[masked] = (np.array(r), np.array(st), np.array(com), np.array(cst), np.array(cen), np.array(rf), np.array(stf), fitted,
                       cur.stat("rmsd_fast_frames"), cur.stat("rmsd_exact_redos"))
        if masked:
            bad = [5]
            assert [f for f in range(nf) if st[f] != 0] == bad, st                           # (the NaN of frame 4 sits in an UNSELECTED atom)
            assert [f for f in range(nf) if cst[f] != 0] == bad
            assert cur.stat("rmsd_fast_frames") > 0                                          # the masked pass did run
            with O.acc64():
                ro4 = O.calc_rmsd(ref_pos, masses, idx, box, frames[4], masses, idx, box)[0]
                assert abs(float(r[4]) - ro4) <= 1e-5 and np.abs(com[4] - O.get_center(frames[4], idx, box, mass=masses)).max() <= 1e-5
                for f in (0, 1, 2, 3, 7):
                    ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box)
                    assert abs(float(r[f]) - ro) <= 1e-5 and abs(float(rf[f]) - ro) <= 1e-5, (f, float(r[f]), float(rf[f]), ro)
                    fin = np.isfinite(frames[f][:, 0])
                    assert np.abs(fitted[f][fin] - want[fin]).max() <= 5e-5, f
                    wc = O.get_center(frames[f], idx, box, mass=masses)
                    assert np.abs(com[f] - wc).max() <= 1e-5, (f, com[f], wc)
            assert np.array_equal(np.nan_to_num(fitted[5], nan=-1.0), np.nan_to_num(frames[5], nan=-1.0))       # a failed frame is left alone
        plan.close(); ref.close(); cur.close()
    a, b = res[1], res[0]
    assert b[8] == 0                                                                          # (the gather run never took the pass)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and np.array_equal(a[6], b[6])
    good = a[1] == 0
    assert np.abs(a[0][good] - b[0][good]).max() <= 2e-6 and np.abs(a[5][good] - b[5][good]).max() <= 2e-6
    assert np.abs(a[2][good] - b[2][good]).max() <= 5e-6 and np.abs(a[4][good] - b[4][good]).max() <= 5e-6
    for f in np.nonzero(good)[0]:
        fin = np.isfinite(a[7][f][:, 0])
        assert np.abs(a[7][f][fin] - b[7][f][fin]).max() <= 3e-5, f


def test_sparse_selections_and_rigid_copies(G):
    """a selection thinner than an eighth of its span keeps to its index list; a rigid copy of the reference through the masked pass is
    handed to the exact-product pass and comes out at rmsd <= 1e-4"""
    n, nf = 80_000, 4
    box = W.box_from_lengths_angles([9.0, 9.0, 9.0], [90.0] * 3)
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.set_tuning(rmsd_fast_min=0)
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.0, W.SEED)                                              # rigid copies
    ref = G.System(n, masses=masses, box=box, positions=cur.get_positions(nf))
    sparse = [(i, i) for i in range(0, n, 10)]
    dense = [(i, i + 1) for i in range(0, n, 4)]
    for s_ in (ref, cur):
        s_.group_create_from_ranges("Sparse", sparse)
        s_.group_create_from_ranges("Dense", dense)
    for name, expect_fast in (("Sparse", False), ("Dense", True)):
        plan = G.RMSDPlan(ref, cur, name)
        f0, r0 = cur.stat("rmsd_fast_frames"), cur.stat("rmsd_exact_redos")
        r, st = plan.rmsd(0, nf)
        assert (st == 0).all() and float(np.max(r)) <= 1e-4, r
        took = (cur.stat("rmsd_fast_frames") - f0) + (cur.stat("rmsd_exact_redos") - r0)
        assert (took == nf) == expect_fast, (name, took)
        if expect_fast:
            assert cur.stat("rmsd_exact_redos") - r0 == nf                                    # rigid copies: every frame handed back
        plan.close()
    ref.close(); cur.close()


@pytest.mark.parametrize("start", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("stride", [2, 3])
def test_unselected_atoms_of_the_span_never_reach_the_image_proof(G, start, stride):
    """The selected atoms are a compact blob, the UNSELECTED atoms between them lie all over the cell: if one of them reached the
    moments or extents of the image proof (as atom 0 of every lane did in the first build of the masked pass -- its mass was zeroed,
    its coordinates were not), the proof would fail (fallbacks > 0) or pick another periodic copy of the centre.  Every residue of
    the first atom mod 4, so that every position of a lane's four atoms is masked somewhere."""
    n, nf = 40_000, 6
    box = W.box_from_lengths_angles([9.0, 8.0, 7.0], [90.0] * 3)
    masses = W.masses_cycle(n)
    blocks = [(i, i) for i in range(start, n - 3, stride)]
    idx = _idx(blocks)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.set_tuning(rmsd_fast_min=0)
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    ref_pos = cur.get_positions(nf)
    rng = np.random.default_rng(start * 10 + stride)
    unsel = np.setdiff1d(np.arange(n), idx)
    frames = []
    for f in range(nf):
        p = cur.get_positions(f)
        p[unsel] = (rng.random((len(unsel), 3)) * np.array([9.0, 8.0, 7.0])).astype(np.float32)
        cur.set_frame(p, box, slot=f)
        frames.append(p)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    for s_ in (ref, cur):
        s_.group_create_from_ranges("S", blocks)
    plan = G.RMSDPlan(ref, cur, "S")
    f0 = cur.stat("rmsd_fast_frames")
    r, st = plan.rmsd(0, nf)
    assert (st == 0).all() and plan.last_fallbacks() == 0 and cur.stat("rmsd_fast_frames") - f0 == nf
    com, _ = cur.group_get_com_batch("S", 0, nf)
    cen, _ = cur.group_get_center_batch("S", 0, nf)
    with O.acc64():
        for f in range(nf):
            assert abs(float(r[f]) - O.calc_rmsd(ref_pos, masses, idx, box, frames[f], masses, idx, box)[0]) <= 1e-5
            assert np.abs(com[f] - O.get_center(frames[f], idx, box, mass=masses)).max() <= 1e-5, f
            assert np.abs(cen[f] - O.get_center(frames[f], idx, box)).max() <= 1e-5, f
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("shape", ["every third atom", "two blocks"])
@pytest.mark.parametrize("cell,streams", [("ortho", 1), ("dodeca", 3)])
def test_rmsd_fit_of_a_masked_selection_through_the_resident_pass(G, shape, cell, streams):
    """calc_rmsd_and_fit of a dense scattered selection in ONE pass (k_fit_resident, the lanes' membership flags carry the mask bits):
    against the oracle and against the two passes on the same frames; unselected atoms all over the cell, a NaN in one of them, a NaN in a
    selected atom, a frame whose image proof fails."""
    n, nf = 70_001, 20
    box = W.box_from_lengths_angles([9.0, 8.0, 7.0], [90.0] * 3) if cell == "ortho" else W.c4_box(9.0)
    masses = W.masses_cycle(n)
    blocks = SHAPES[shape](n)
    idx = _idx(blocks)
    unsel = np.setdiff1d(np.arange(n), idx)
    rng = np.random.default_rng(5)
    res = {}
    for resident in (2, 0):
        cur = G.System(n, masses=masses, n_slots=nf + 1)
        cur.set_tuning(resident=resident, resident_streams=streams)
        cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
        cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
        ref_pos = cur.get_positions(nf)
        frames = [cur.get_positions(f) for f in range(nf)]
        if resident == 2:
            scatter = (rng.random((len(unsel), 3)) @ W.box_matrix(box)).astype(np.float32)
        for f in (1, 8, 19):
            frames[f] = frames[f].copy(); frames[f][unsel] = scatter                          # unselected atoms anywhere in the cell
        frames[3] = W.proof_failing_frame(ref_pos, box, "two_lobes", 9)
        frames[6] = frames[6].copy(); frames[6][unsel[len(unsel) // 2]] = np.nan
        frames[11] = frames[11].copy(); frames[11][idx[len(idx) // 3]] = np.nan
        ref = G.System(n, masses=masses, box=box, positions=ref_pos)
        for s_ in (ref, cur):
            s_.group_create_from_ranges("S", blocks)
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        plan = G.RMSDPlan(ref, cur, "S")
        cur.profile_enable(True)
        r, st = plan.rmsd_fit(0, nf, raise_on_error=False)
        prof = cur.profile_read()
        assert (prof["k_fit_resident"][1] > 0) == (resident == 2), prof
        fitted = [cur.get_positions(f) for f in range(nf)]
        res[resident] = (np.array(r), np.array(st), fitted)
        if resident == 2:
            assert [f for f in range(nf) if st[f] != 0] == [11], st
            assert plan.last_fallbacks() >= 1                                                 # frame 3 went back to the exact path
            with O.acc64():
                for f in (0, 1, 3, 4, 8, 12, 19):
                    ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box)
                    assert abs(float(r[f]) - ro) <= 1e-5, (f, float(r[f]), ro)
                    assert np.abs(fitted[f] - want).max() <= 5e-5, f
            assert np.array_equal(np.nan_to_num(fitted[11], nan=-1.0), np.nan_to_num(frames[11], nan=-1.0))     # a failed frame is left alone
        plan.close(); ref.close(); cur.close()
    a, b = res[2], res[0]
    assert np.array_equal(a[1], b[1])
    good = a[1] == 0
    assert np.abs(a[0][good] - b[0][good]).max() <= 2e-6
    for f in np.nonzero(good)[0]:
        fin = np.isfinite(a[2][f][:, 0]) & np.isfinite(b[2][f][:, 0])
        assert np.array_equal(np.isfinite(a[2][f][:, 0]), np.isfinite(b[2][f][:, 0]))
        assert np.abs(a[2][f][fin] - b[2][f][fin]).max() <= 3e-5, f


@pytest.mark.parametrize("shape", ["nine in ten", "pairs", "two blocks"])
def test_translate_and_wrap_of_a_masked_group_are_the_list_paths_bit_for_bit(G, shape):
    """group translate / wrap of a scattered selection that covers at least half of its span walk the span with the mask
    (k_translate_wrap); every atom goes through the same arithmetic as on the index-list path, so the frames must come out
    identical bit for bit, atoms outside the selection untouched, an atom without position named the same way"""
    n, nf = 50_001, 5
    box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [75.0, 80.0, 70.0])
    blocks = {"nine in ten": [(i, i + 8) for i in range(3, n - 10, 10)], "pairs": [(i, i + 1) for i in range(1, n - 2, 4)], "two blocks": [(5, n // 3), (n // 2 + 3, n - 7)]}[shape]
    idx = _idx(blocks)
    out = {}
    for masked in (1, 0):
        cur = G.System(n, masses=W.masses_cycle(n), n_slots=nf + 1)
        cur.set_tuning(masked_selections=masked)
        cur.synth_reference(nf, box, 0.45 * min(box[:3]), W.SEED)
        cur.synth_frames(nf, 0, nf, 0, 0.3, W.SEED)
        before = [cur.get_positions(f) for f in range(nf)]
        before[3] = before[3].copy(); before[3][idx[len(idx) // 2]] = np.nan
        cur.set_frame(before[3], box, slot=3)
        cur.group_create_from_ranges("S", blocks)
        st1 = cur.group_translate_batch("S", [3.3, -7.1, 0.4], 0, nf, raise_on_error=False)
        mid = [cur.get_positions(f) for f in range(nf)]
        st2 = cur.group_wrap_batch("S", 0, nf, raise_on_error=False)
        out[masked] = (np.array(st1), np.array(st2), mid, [cur.get_positions(f) for f in range(nf)])
        if masked:
            other = np.setdiff1d(np.arange(n), idx)
            for f in range(nf):
                assert np.array_equal(mid[f][other], before[f][other]) and np.array_equal(out[1][3][f][other], before[f][other])
            assert st1[3] != 0 and all(st1[f] == 0 for f in (0, 1, 2, 4))
        cur.close()
    a, b = out[1], out[0]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for f in range(nf):
        assert np.array_equal(a[2][f], b[2][f], equal_nan=True) and np.array_equal(a[3][f], b[3][f], equal_nan=True), f
