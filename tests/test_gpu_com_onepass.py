"""GPU parity of the one-pass get_center / get_com (the RMSD sums pass without a reference, groan_rs_amd/csrc/gr_kernels.h
k_rmsd_accum<0, true, true>) against the oracle's restatement of the reference's two dependent passes
(src/structures/iterators.rs:1237-1266, 1404-1438: unweighted Bai-Breen estimate, then the mean of c' + vector_to(c', x)), and
against the library's own two-pass kernels.  The result is NOT wrapped by the reference: it lies in the periodic copy that
the estimate c' (inside the cell) selects, so a group drifting through a cell face must come out on the right side."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def cluster(rng, n, box, centre_frac, sigma):
    boxm = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    c = np.asarray(centre_frac, np.float64) @ boxm
    return (c + rng.normal(0, sigma, (n, 3))).astype(np.float32)


BOXES = {"ortho": ([9.0, 8.0, 7.0], [90.0, 90.0, 90.0]), "dodecahedron": ([9.0, 9.0, 9.0], [60.0, 60.0, 90.0]),
         "octahedron": ([9.0, 9.0, 9.0], [70.53, 109.47, 70.53]), "triclinic": ([9.5, 9.0, 8.5], [75.0, 80.0, 70.0])}


@pytest.mark.parametrize("bname", list(BOXES))
@pytest.mark.parametrize("weighted", [True, False])
def test_cluster_anywhere_in_the_cell_matches_the_oracle(G, bname, weighted):
    box = O.box_from_lengths_angles(*BOXES[bname])
    rng = np.random.default_rng(hash(bname) % 1000)
    n, nsel = 60_000, 50_001
    masses = np.array([1.008, 12.011, 14.007, 15.999, 32.06], np.float32)[np.arange(n) % 5]
    idx = np.arange(777, 777 + nsel)
    s = G.System(n, masses=masses, n_slots=1)
    s.group_create_from_ranges("G", [(777, 777 + nsel - 1)])
    fb0 = s.center_fallbacks()
    # centres: mid cell, near faces / edges / a corner (the copy must follow c'), broken across the boundary by wrapping
    for cf in ([0.5, 0.5, 0.5], [0.03, 0.5, 0.5], [0.97, 0.04, 0.5], [0.02, 0.98, 0.03], [0.25, 0.75, 0.99]):
        pos = cluster(rng, n, box, cf, 0.3)
        pos = O.wrap_atoms(pos, np.arange(n), box)            # a compact group, scattered over the periodic copies
        s.set_frame(pos, box)
        got = np.array(s.group_get_com("G") if weighted else s.group_get_center("G"))
        with O.acc64():
            want = O.get_center(pos, idx, box, mass=masses if weighted else None)
        assert np.abs(got - want).max() <= TOL, (bname, cf, got, want)
        s.set_center_onepass_min(0)                           # the library's own two passes: same answer
        two = np.array(s.group_get_com("G") if weighted else s.group_get_center("G"))
        s.set_center_onepass_min(4096)
        assert np.abs(got - two).max() <= TOL
    assert s.center_fallbacks() == fb0                        # none of these needed the two passes
    s.close()


def test_frames_the_proof_rejects_take_the_two_passes(G):
    """a group wider than half the box, and a compact one whose centre sits on a cell face (the periodic copy then depends
    on which side c' falls): both must come back with the reference's answer through the fallback"""
    box = np.array([8.0, 7.0, 6.0, 0, 0, 0, 0, 0, 0], np.float32)
    rng = np.random.default_rng(9)
    n = 40_000
    masses = np.array([1.008, 12.011, 15.999], np.float32)[np.arange(n) % 3]
    s = G.System(n, masses=masses, n_slots=1)
    s.group_create_from_ranges("G", [(0, n - 1)])
    idx = np.arange(n)
    wide = (rng.random((n, 3)) * box[:3] * [0.8, 1.0, 0.3]).astype(np.float32)
    face = O.wrap_atoms(cluster(rng, n, box, [1.0 - 2e-5, 0.5, 0.5], 0.3), idx, box)
    for pos in (wide, face):
        fb0 = s.center_fallbacks()
        s.set_frame(pos, box)
        got = np.array(s.group_get_com("G"))
        assert s.center_fallbacks() == fb0 + 1
        with O.acc64():
            want = O.get_center(pos, idx, box, mass=masses)
        assert np.abs(got - want).max() <= 2e-5, (got, want)
    s.close()


def test_errors_in_reference_order_and_batches(G):
    """positions of the whole group before any mass (iterators.rs:1405-1422); per-frame results in a batch, one of them
    falling back, one failing"""
    box = O.box_from_lengths_angles([9.0, 9.0, 9.0], [60.0, 60.0, 90.0])
    rng = np.random.default_rng(12)
    n, nf = 30_000, 6
    masses = np.array([1.008, 12.011, 15.999], np.float32)[np.arange(n) % 3]
    s = G.System(n, masses=masses, n_slots=nf)
    s.group_create_from_ranges("G", [(100, n - 101)])
    idx = np.arange(100, n - 100)
    frames = []
    for f in range(nf):
        pos = O.wrap_atoms(cluster(rng, n, box, rng.random(3), 0.3), np.arange(n), box)
        frames.append(pos)
    frames[2] = (rng.random((n, 3)) * 9.0).astype(np.float32)                 # wide: falls back
    bad = frames[4].copy(); bad[20_000, 0] = np.nan; frames[4] = bad          # fails
    for f in range(nf):
        s.set_frame(frames[f], box, slot=f)
    out, status = s.group_get_com_batch("G", 0, nf, raise_on_error=False)
    assert status[4] != 0 and all(status[f] == 0 for f in (0, 1, 2, 3, 5)) and np.isnan(out[4]).all()
    with O.acc64():
        for f in (0, 1, 2, 3, 5):
            want = O.get_center(frames[f], idx, box, mass=masses)
            assert np.abs(out[f] - want).max() <= 2e-5, (f, out[f], want)
    with pytest.raises(G.GroupError) as e:
        s.group_get_com("G", slot=4)
    assert e.value.variant == "InvalidPosition" and e.value.detail == 20_000
    m2 = masses.copy(); m2[150] = np.nan
    s2 = G.System(n, masses=m2, n_slots=1)
    s2.group_create_from_ranges("G", [(100, n - 101)])
    s2.set_frame(bad, box)
    with pytest.raises(G.GroupError) as e:
        s2.group_get_com("G")
    assert e.value.variant == "InvalidPosition" and e.value.detail == 20_000   # the position error wins over the earlier atom's mass
    s2.set_frame(frames[0], box)
    with pytest.raises(G.GroupError) as e:
        s2.group_get_com("G")
    assert e.value.variant == "InvalidMass" and e.value.detail == 150
    assert np.abs(np.array(s2.group_get_center("G")) - O.get_center(frames[0], idx, box)).max() <= 1e-4   # unweighted: no masses needed (f32 oracle sums)
    s.close(); s2.close()


@pytest.mark.parametrize("bname", list(BOXES))
def test_centre_on_a_cell_face_takes_the_copy_the_estimate_selects(G, bname):
    """the group's centre within the proof's bound of one, two or three cell faces: the images are proven, the periodic copy is
    not -- the masked estimate pass selects it (GR_ST_AMBIG); the answer must be the reference's, whichever side c' falls"""
    box = O.box_from_lengths_angles(*BOXES[bname])
    rng = np.random.default_rng(31)
    n = 20_000
    masses = np.array([1.008, 12.011, 15.999], np.float32)[np.arange(n) % 3]
    s = G.System(n, masses=masses, n_slots=1)
    s.group_create_from_ranges("G", [(0, n - 1)])
    idx = np.arange(n)
    for cf in ([1.0 - 1e-5, 0.5, 0.5], [1e-5, 1.0 - 2e-5, 0.5], [3e-6, 1.0 - 3e-6, 1e-6], [0.0, 0.0, 0.0], [0.5, 0.99999, 0.00001]):
        raw = cluster(rng, n, box, cf, 0.25).astype(np.float64)
        boxm = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
        raw += np.asarray(cf, np.float64) @ boxm - raw.mean(axis=0)               # the SAMPLE mean on the face (the proof's radius is ~3e-4 cell widths
        pos = O.wrap_atoms(raw.astype(np.float32), idx, box)                      # since round 4's third-moment bound: a sample mean 5e-4 off is no longer ambiguous)
        fb0 = s.center_fallbacks()
        s.set_frame(pos, box)
        for weighted in (True, False):
            got = np.array(s.group_get_com("G") if weighted else s.group_get_center("G"))
            with O.acc64():
                want = O.get_center(pos, idx, box, mass=masses if weighted else None)
            assert np.abs(got - want).max() <= 2e-5, (bname, cf, weighted, got, want)
        assert s.center_fallbacks() == fb0 + 2
    s.close()
