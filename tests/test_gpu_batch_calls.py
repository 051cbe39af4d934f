"""Batched per-frame analyses (gr_group_center_batch / gr_atoms_center_batch / gr_group_wrap_batch / gr_group_translate_batch):
the same results, bit for bit, as the per-frame calls (System::group_get_com / group_estimate_com / atoms_center_mass /
atoms_wrap / group_translate, src/system/analysis.rs:52-320, utility.rs:109-185, modifying.rs:45-75,201-222) -- with the
oracle as the referee on a sample -- and per-frame error reporting."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def make(G, n, nf, box, seed, scattered=True):
    rng = np.random.default_rng(seed)
    masses = rng.uniform(1.0, 16.0, n).astype(np.float32)
    s = G.System(n, masses=masses, n_slots=nf)
    L = np.array([box[0], box[1], box[2]], np.float32)
    frames = []
    for f in range(nf):
        blob = rng.normal(0, 0.6, (n, 3)) + rng.uniform(0, 1, 3) * L
        pos = O.wrap_atoms(blob.astype(np.float32), np.arange(n), box)
        s.set_frame(pos, box, slot=f); frames.append(pos)
    s.group_create_from_ranges("A", [(10, n // 2)])
    idxB = np.unique(rng.integers(0, n, n // 5))
    s.group_create_from_indices("B", idxB)
    return s, frames, masses, idxB


@pytest.mark.parametrize("angles", [[90.0, 90.0, 90.0], [60.0, 60.0, 90.0]])
def test_center_batch_equals_per_frame_calls_and_oracle(G, angles):
    box = O.box_from_lengths_angles([7.0, 7.0, 7.0], angles)
    n, nf = 40_000, 9
    s, frames, m, idxB = make(G, n, nf, box, 3)
    for grp, idx in (("A", np.arange(10, n // 2 + 1)), ("B", idxB), ("all", np.arange(n))):
        for fn_b, fn_1, orc, w in ((s.group_get_com_batch, s.group_get_com, O.get_center, True), (s.group_get_center_batch, s.group_get_center, O.get_center, False),
                                   (s.group_estimate_com_batch, s.group_estimate_com, O.estimate_center, True)):
            got, st = fn_b(grp, 0, nf)
            assert (st == 0).all()
            for f in range(nf):
                assert np.array_equal(got[f], fn_1(grp, slot=f))
            with O.acc64():
                want = orc(frames[2], idx, box, mass=m if w else None)
            assert np.abs(got[2] - want).max() <= 1e-5
    s.close()


def test_center_wrap_translate_batch(G):
    box = np.array([6.0, 7.0, 8.0, 0, 0, 0, 0, 0, 0], np.float32)
    n, nf = 30_000, 7
    s, frames, m, idxB = make(G, n, nf, box, 8)
    t, _, _, _ = make(G, n, nf, box, 8)                                   # identical twin driven by the per-frame calls
    assert (s.atoms_center_batch("B", 0, nf, G.Dimension.XYZ, weighted=True) == 0).all()
    for f in range(nf):
        t.atoms_center_mass("B", G.Dimension.XYZ, slot=f)
        assert np.array_equal(s.get_positions(f), t.get_positions(f))
    assert (s.atoms_center_batch("A", 0, nf, G.Dimension.XZ) == 0).all()
    for f in range(nf):
        t.atoms_center("A", G.Dimension.XZ, slot=f)
        assert np.array_equal(s.get_positions(f), t.get_positions(f))
    assert (s.group_translate_batch("B", [3.3, -9.1, 0.4], 0, nf) == 0).all()
    assert (s.group_wrap_batch(None, 0, nf) == 0).all()
    for f in range(nf):
        t.group_translate("B", [3.3, -9.1, 0.4], slot=f); t.atoms_wrap(slot=f)
        assert np.array_equal(s.get_positions(f), t.get_positions(f))
    want = O.atoms_center(frames[1], idxB, "xyz", box, mass=m)
    s.close(); t.close()
    assert want.shape == (n, 3)


def test_every_frame_is_judged_on_its_own(G):
    box = np.array([6.0, 7.0, 8.0, 0, 0, 0, 0, 0, 0], np.float32)
    n, nf = 5_000, 6
    s, frames, m, idxB = make(G, n, nf, box, 21)
    s.reset_box(slot=2)                                                    # frame 2: no box
    bad = frames[4].copy(); bad[int(idxB[7]), 0] = np.nan                  # frame 4: an atom of B without position
    s.set_frame(bad, box, slot=4)
    got, st = s.group_get_com_batch("B", 0, nf, raise_on_error=False)
    assert st.tolist() == [0, 0, G._lib.E_NO_BOX, 0, G._lib.E_NO_POSITION, 0]
    assert np.isnan(got[2]).all() and np.isnan(got[4]).all() and np.isfinite(got[[0, 1, 3, 5]]).all()
    with pytest.raises(G.GroupError) as e:                                 # the first failing frame's error, like the loop in the reference would raise
        s.group_get_com_batch("B", 0, nf)
    assert e.value.variant == "InvalidSimBox"
    before = [s.get_positions(f) for f in range(nf)]
    st = s.atoms_center_batch("B", 0, nf, G.Dimension.XYZ, raise_on_error=False)
    assert st.tolist() == [0, 0, G._lib.E_NO_BOX, 0, G._lib.E_NO_POSITION, 0]
    for f in (2, 4):                                                       # failed frames are left untouched
        a, b = s.get_positions(f), before[f]
        assert np.array_equal(a[~np.isnan(b)], b[~np.isnan(b)])
    assert not np.array_equal(s.get_positions(0), before[0])
    with pytest.raises(G.GroupError):
        s.group_get_com_batch("nope", 0, nf)
    s.close()


def test_wrap_values_where_rounding_decides(G):
    """atoms_wrap / atoms_translate against the reference's loops (vector3d.rs:398-417) bit for bit on tiny negatives (the loop's
    `w += L` rounds to exactly L and stops there: the closed upper end), exact multiples of L and the neighbours of 0 and L"""
    L = np.array([6.5, 7.25, 3.0], np.float32)
    box = np.array([L[0], L[1], L[2], 0, 0, 0, 0, 0, 0], np.float32)
    vals = []
    for m in (-1, 0, 1, 2):
        for k in range(-3, 4):
            t = (np.float32(m) * L).astype(np.float32)
            for _ in range(abs(k)):
                t = np.nextafter(t, np.float32(np.inf if k > 0 else -np.inf)).astype(np.float32)
            vals.append(t)
    for e in (1e-9, 1e-8, 3e-7, 1e-12, 1e-30, 1e-45):
        vals += [np.full(3, -e, np.float32), np.full(3, e, np.float32), (L - np.float32(e)).astype(np.float32), (L + np.float32(e)).astype(np.float32)]
    pos = np.array(vals, np.float32)
    n = pos.shape[0]
    s = G.System(n, n_slots=1)
    s.set_frame(pos, box)
    s.atoms_wrap()
    got = s.get_positions()
    want = O.wrap_atoms(pos, np.arange(n), box)
    assert np.array_equal(got.view(np.uint32) & 0x7fffffff, want.view(np.uint32) & 0x7fffffff), np.argwhere(got != want)[:5]   # (the sign of a zero is not compared)
    s.set_frame(pos, box)
    s.atoms_translate([1e-9, -1e-9, 0.0])
    got = s.get_positions()
    want = O.translate(pos, np.arange(n), [1e-9, -1e-9, 0.0], box)
    assert np.array_equal(got.view(np.uint32) & 0x7fffffff, want.view(np.uint32) & 0x7fffffff)
    s.close()
