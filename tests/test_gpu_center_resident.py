"""atoms_center / atoms_center_mass of the whole system as ONE pass over HBM (gr_resident.h MODE 1, GR_TUNE_CENTER_RESIDENT; utility.rs:109-185):
the same bits as the two passes it replaces (centre estimate, then translate + wrap), the oracle as referee on a sample, and the frames it must
hand back -- an atom without position, a launch that never starts, a launch aborted half way."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def build(G, n, nf, box, seed, spread=0.9):
    rng = np.random.default_rng(seed)
    masses = rng.uniform(1.0, 16.0, n).astype(np.float32)
    s = G.System(n, masses=masses, n_slots=nf)
    L = np.array([box[0], box[1], box[2]], np.float32)
    frames = []
    for f in range(nf):
        # a blob that hangs over the cell's faces (so the wrap has work to do), a few atoms several cells away (the general wrap's turns)
        blob = rng.normal(0, spread, (n, 3)) + rng.uniform(0, 1, 3) * L
        far = rng.integers(0, n, 5)
        blob[far] += rng.integers(-3, 4, (5, 3)) * L
        pos = blob.astype(np.float32)
        s.set_frame(pos, box, slot=f); frames.append(pos)
    return s, frames, masses


def both_ways(G, s, frames, box, group, nf, dim, weighted, **tuning):
    """-> (positions after the resident form, positions after the two passes, statuses of both)"""
    res = {}
    for mode in (1, 0):
        for f in range(nf):
            s.set_frame(frames[f], box, slot=f)
        s.set_tuning(center_resident=mode, **tuning)
        st = s.atoms_center_batch(group, 0, nf, dim, weighted=weighted, raise_on_error=False)
        res[mode] = ([s.get_positions(f) for f in range(nf)], np.array(st))
    return res


@pytest.mark.parametrize("angles", [[90.0, 90.0, 90.0], [60.0, 60.0, 90.0], [75.0, 80.0, 70.0]])
@pytest.mark.parametrize("n,streams", [(20_011, 0), (150_000, 3)])
def test_same_bits_as_the_two_passes(G, angles, n, streams):
    box = O.box_from_lengths_angles([9.0, 8.5, 8.0], angles)
    nf = 21
    s, frames, m = build(G, n, nf, box, 5)
    s.group_create_from_ranges("most", [(7, n - 1 - n // 3)])
    for group, dim, weighted in (("all", G.Dimension.XYZ, True), ("all", G.Dimension.XYZ, False), ("most", G.Dimension.XZ, True), ("most", G.Dimension.Y, False)):
        before = s.stat("center_res_launches")
        res = both_ways(G, s, frames, box, group, nf, dim, weighted, resident=2, resident_streams=streams)
        assert s.stat("center_res_launches") == before + 1, "the resident form was not taken"
        assert (res[1][1] == 0).all() and (res[0][1] == 0).all()
        for f in range(nf):
            assert np.array_equal(res[1][0][f], res[0][0][f]), (group, int(dim), weighted, f, np.abs(res[1][0][f] - res[0][0][f]).max())
    # the oracle on one frame (the reference requires an orthogonal cell)
    if angles == [90.0, 90.0, 90.0]:
        want = O.atoms_center(frames[3], np.arange(n), "xyz", box, mass=m)
        s.set_tuning(center_resident=1, resident=2)
        for f in range(nf):
            s.set_frame(frames[f], box, slot=f)
        s.atoms_center_batch("all", 0, nf, G.Dimension.XYZ, weighted=True)
        assert np.abs(s.get_positions(3) - want).max() < 2e-5
    s.close()


def test_frames_it_hands_back(G):
    """an atom without position inside / outside the reference group, a frame without a box: the launch leaves those frames alone and the two
    passes report them exactly as they always did; every other frame of the batch is moved by the launch"""
    box = O.box_from_lengths_angles([9.0, 8.5, 8.0], [90.0, 90.0, 90.0])
    n, nf = 60_000, 20
    s, frames, m = build(G, n, nf, box, 9)
    s.group_create_from_ranges("half", [(0, n // 2)])
    frames[4] = frames[4].copy(); frames[4][100] = np.nan             # inside the group
    frames[11] = frames[11].copy(); frames[11][n - 5] = np.nan        # outside it
    res = {}
    for mode in (1, 0):
        for f in range(nf):
            s.set_frame(frames[f], box, slot=f)
        s.reset_box(slot=7)                                           # no box
        s.set_tuning(center_resident=mode, resident=2)
        before = s.stat("center_res_redone")
        st = s.atoms_center_batch("half", 0, nf, G.Dimension.XYZ, weighted=True, raise_on_error=False)
        res[mode] = ([s.get_positions(f) for f in range(nf)], np.array(st), s.stat("center_res_redone") - before)
    assert np.array_equal(res[1][1], res[0][1]), (res[1][1], res[0][1])
    assert res[1][1][4] != 0 and res[1][1][11] != 0 and res[1][1][7] != 0 and (np.delete(res[1][1], [4, 7, 11]) == 0).all()
    assert res[1][2] == 2 and res[0][2] == 0
    for f in range(nf):
        assert np.array_equal(res[1][0][f], res[0][0][f], equal_nan=True), f
    s.close()


def test_a_launch_that_never_starts_and_one_that_is_aborted(G):
    box = O.box_from_lengths_angles([9.0, 8.5, 8.0], [60.0, 60.0, 90.0])
    n, nf = 60_000, 40
    s, frames, m = build(G, n, nf, box, 13)
    want = both_ways(G, s, frames, box, "all", nf, G.Dimension.XYZ, True, resident=2)[0]
    # never started: every frame goes through the two passes
    for f in range(nf):
        s.set_frame(frames[f], box, slot=f)
    s.set_tuning(center_resident=1, resident=2, test_resident_no_start=1)
    misses = s.stat("res_handshake_misses")
    st = s.atoms_center_batch("all", 0, nf, G.Dimension.XYZ, weighted=True, raise_on_error=False)
    assert s.stat("res_handshake_misses") == misses + 1 and (np.array(st) == 0).all()
    for f in range(nf):
        assert np.array_equal(s.get_positions(f), want[0][f]), f
    # aborted at frame 17: frames before it are the launch's, the untouched ones are redone; a frame caught half-moved is reported, never silent
    for f in range(nf):
        s.set_frame(frames[f], box, slot=f)
    s.set_tuning(center_resident=1, resident=2, test_resident_abort_at=17, resident_streams=1)
    aborts = s.stat("res_aborts")
    for _ in range(8):      # (the backoff after the miss above makes the context sit out a few batches)
        st = np.array(s.atoms_center_batch("all", 0, nf, G.Dimension.XYZ, weighted=True, raise_on_error=False))
        if s.stat("res_aborts") > aborts:
            break
        for f in range(nf):
            s.set_frame(frames[f], box, slot=f)
        s.set_tuning(test_resident_abort_at=17)
    assert s.stat("res_aborts") == aborts + 1
    for f in range(nf):
        if st[f] == 0:
            assert np.array_equal(s.get_positions(f), want[0][f]), f
    assert (st == 0).sum() >= nf - 8
    s.close()


def test_the_switch_takes_zero_or_one(G):
    s = G.System(100)
    s.set_tuning(center_resident=0); s.set_tuning(center_resident=1)
    with pytest.raises(G.DeviceError):
        s.set_tuning(center_resident=2)
    s.close()
