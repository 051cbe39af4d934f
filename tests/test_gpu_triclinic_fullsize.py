"""GPU parity for the triclinic extension (PARITY UNPINNED by the reference: validated against the oracle's
definition + an fp64 brute-force image search) and full-size (1e6-atom) checks through size-independent
properties: fit -> refit idempotence, rigid-motion + PBC-break invariance, COM restoration, matrix symmetry."""
import itertools

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
SEED = 20260424


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def lattice(box9):
    b = np.asarray(box9, np.float64)
    return np.array([[b[0], 0, 0], [b[5], b[1], 0], [b[7], b[8], b[2]]])


def brute_dist(a, b, box9, rng=4):
    L = lattice(box9)
    ks = np.array(list(itertools.product(range(-rng, rng + 1), repeat=3)), np.float64)
    d = (a.astype(np.float64) - b.astype(np.float64))[None, :] + ks @ L
    return np.sqrt((d ** 2).sum(1).min())


@pytest.mark.parametrize("name", ["triclinic", "octahedron", "dodecahedron"])
def test_small_nonorthogonal_fixtures(G, tric_small, name):
    frames, boxes = tric_small[name + "_frames"], tric_small[name + "_boxes9"]
    n = frames.shape[1]
    m = (np.arange(n) % 5 + 1).astype(np.float32)
    s = G.System(n, masses=m, n_slots=2)
    s.group_create_from_ranges("A", [(0, 19)]); s.group_create_from_indices("B", list(range(20, n, 2)))
    iA, iB = np.arange(20), np.arange(20, n, 2)
    for f in (0, frames.shape[0] - 1):
        pos, box = frames[f], boxes[f]
        s.set_frame(pos, box)
        for grp, idx in (("A", iA), ("B", iB), ("all", np.arange(n))):
            np.testing.assert_allclose(s.group_estimate_center(grp), O.estimate_center(pos, idx, box), atol=TOL, rtol=0)
            np.testing.assert_allclose(s.group_estimate_com(grp), O.estimate_center(pos, idx, box, mass=m), atol=TOL, rtol=0)
        for dim in ("XYZ", "X", "YZ"):
            got = s.group_all_distances("A", "B", G.Dimension[dim])
            np.testing.assert_allclose(got, O.group_all_distances(pos, iA, iB, dim.lower(), box), atol=3e-6, rtol=0)
        got = s.group_all_distances("A", "B")
        for i, j in ((0, 0), (3, 7), (19, 14), (11, 2)):
            assert abs(got[i, j] - brute_dist(pos[iA[i]], pos[iB[j]], box)) <= 2e-5     # the true global minimum image
        s.atoms_translate([7.7, -3.1, 12.9])
        np.testing.assert_allclose(s.get_positions(), O.translate(pos, np.arange(n), [7.7, -3.1, 12.9], box), atol=TOL, rtol=0)
        s.set_frame(pos, box)
        s.atoms_center("A", G.Dimension.XYZ)
        np.testing.assert_allclose(s.get_positions(), O.atoms_center(pos, iA, "xyz", box), atol=TOL, rtol=0)
        # strict mode = the reference's behaviour (system/mod.rs:1120-1124)
        s.set_strict_orthogonal(True)
        with pytest.raises(G.GroupError) as e:
            s.group_get_center("A")
        assert e.value.variant == "InvalidSimBox" and e.value.detail.variant == "NotOrthogonal"
        s.set_strict_orthogonal(False)
    s.close()


def make_blob_system(G, n, box, radius, n_frames, noise=0.05):
    masses = np.array([1.008, 12.011, 14.007, 15.999], np.float32)[np.arange(n) % 4]
    cur = G.System(n, masses=masses, n_slots=n_frames + 1)
    cur.synth_reference(n_frames, box, radius, SEED)
    cur.synth_frames(n_frames, 0, n_frames, 0, noise, SEED)
    ref_pos = cur.get_positions(n_frames)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    return ref, cur, masses, ref_pos


@pytest.mark.parametrize("lengths,angles", [([6.0, 6.0, 6.0], [60.0, 60.0, 90.0]), ([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]),
                                            ([6.0, 6.0, 6.0], [70.53, 109.47, 70.53]), ([6.0, 7.0, 5.0], [90.0, 90.0, 90.0])])
def test_blob_rmsd_fit_single_pass_vs_exact_vs_oracle(G, lengths, angles):
    box = O.box_from_lengths_angles(lengths, angles)
    n, nf = 50_000, 3
    ref, cur, masses, ref_pos = make_blob_system(G, n, box, 0.8, nf)
    frames = [cur.get_positions(f) for f in range(nf)]
    idx = np.arange(n)
    plan = G.RMSDPlan(ref, cur, "all")
    r, st, R = plan.rmsd(0, nf, return_rotation=True)
    assert (st == 0).all() and plan.last_fallbacks() == 0          # compact blob: the single-pass image proof holds
    plan.force_exact(True)
    r2, st2, R2 = plan.rmsd(0, nf, return_rotation=True)
    assert np.abs(r - r2).max() <= 2e-6 and np.abs(R - R2).max() <= 2e-6
    plan.force_exact(False)
    rf, _ = plan.rmsd_fit(0, nf)
    O.set_accumulate_f64(True)
    try:
        for f in range(nf):
            ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box)
            assert abs(r[f] - ro) <= TOL and abs(rf[f] - ro) <= TOL
            np.testing.assert_allclose(cur.get_positions(f), want, atol=3e-5, rtol=0)
    finally:
        O.set_accumulate_f64(False)
    ref.close(); cur.close()


def test_extended_group_takes_the_exact_path_and_still_matches(G):
    """two lobes 0.6 L apart: wider than half the box, so the single-pass image proof must fail and the
    multi-pass path (Bai-Breen centre -> unwrap -> COM -> accumulate) must reproduce the oracle"""
    box = np.array([8.0, 7.0, 6.0, 0, 0, 0, 0, 0, 0], np.float32)
    rng = np.random.default_rng(4)
    n = 4000
    lobe = rng.normal(0, 0.35, (n, 3))
    lobe[: n // 3, 0] += 4.8                                    # second, smaller lobe across the periodic boundary
    pos = (lobe + [1.0, 3.5, 3.0]).astype(np.float32)
    pos = O.wrap_atoms(pos, np.arange(n), box)
    m = rng.uniform(1, 16, n).astype(np.float32)
    ref = G.System(n, masses=m, box=box, positions=pos)
    cur = G.System(n, masses=m, n_slots=1)
    moved = O.translate(pos + rng.normal(0, 0.02, (n, 3)).astype(np.float32), np.arange(n), [2.2, -1.0, 0.7], box)
    cur.set_frame(moved, box)
    plan = G.RMSDPlan(ref, cur, "all")
    r, st = plan.rmsd(0, 1)
    assert st[0] == 0 and plan.last_fallbacks() == 1
    ro, want = O.calc_rmsd_and_fit(pos, m, np.arange(n), box, moved, m, np.arange(n), box)
    assert abs(r[0] - ro) <= TOL
    rf, _ = plan.rmsd_fit(0, 1)
    np.testing.assert_allclose(cur.get_positions(), want, atol=3e-5, rtol=0)
    np.testing.assert_allclose(ref.group_get_com("all"), O.get_center(pos, np.arange(n), box, mass=m), atol=TOL, rtol=0)
    ref.close(); cur.close()


def test_full_size_properties_1e6_atoms(G):
    """BASELINE-size frames: no oracle (seconds per frame), so size-independent properties instead."""
    n, nf = 1_000_000, 4
    box = O.box_from_lengths_angles([24.18, 24.18, 24.18], [60.0, 60.0, 90.0])
    radius = 0.2 * float(min(box[0], box[1], box[2]))
    ref, cur, masses, ref_pos = make_blob_system(G, n, box, radius, nf, noise=0.05)
    plan = G.RMSDPlan(ref, cur, "all")
    r0, st, R0 = plan.rmsd(0, nf, return_rotation=True)
    assert (st == 0).all() and plan.last_fallbacks() == 0
    # noise = 4 uniforms -> sigma 0.05 per axis -> rmsd ~ 0.05*sqrt(3) for every frame, whatever its pose
    assert np.all(np.abs(r0 - 0.05 * np.sqrt(3.0)) <= 2e-3)
    plan.force_exact(True)
    r_exact, _ = plan.rmsd(0, nf)
    plan.force_exact(False)
    assert np.abs(r0 - r_exact).max() <= 2e-6                      # single-pass == multi-pass at full size
    rf, _ = plan.rmsd_fit(0, nf)
    assert np.abs(rf - r0).max() <= 2e-6   # the fit pass's direct sum vs the closed form of the rmsd-only pass
    # idempotence: a fitted frame is already optimally superposed -> same RMSD, identity rotation
    r1, _, R1 = plan.rmsd(0, nf, return_rotation=True)
    assert np.abs(r1 - r0).max() <= TOL
    assert np.abs(R1 - np.eye(3)[None]).max() <= 2e-5
    # the fitted group sits on the reference: COM restored, every atom within the noise envelope
    com_ref = ref.group_get_com("all")
    for f in range(nf):
        fitted = cur.get_positions(f)
        assert np.abs(cur.group_get_com("all", slot=f) - com_ref).max() <= 2e-5
        d = fitted - ref_pos
        assert np.abs(d).max() <= 0.05 * np.sqrt(3.0) * 2.0 * np.sqrt(3.0) + 0.02   # |noise| <= 2*sqrt(3)*sigma per axis, mixed by R
        assert abs(np.sqrt((d.astype(np.float64) ** 2).sum(1).mean()) - 0.05 * np.sqrt(3.0)) <= 2e-3
    # zero noise: every rigidly moved, PBC-broken copy has rmsd ~ 0 (sqrt of the f32 input rounding)
    cur.synth_frames(nf, 0, nf, 100, 0.0, SEED)
    rz, _ = plan.rmsd_fit(0, nf)
    assert np.all(rz <= 2e-4)
    for f in range(nf):
        assert np.abs(cur.get_positions(f) - ref_pos).max() <= 3e-4
    ref.close(); cur.close()


def test_pair_distances_config3_triclinic(G):
    """config[2]: 1e6 atoms uniform in a triclinic cell, 1e4 x 1e4 min-image distances, whole matrix vs the oracle"""
    n = 1_000_000
    box = O.box_from_lengths_angles([24.0, 23.0, 22.0], [75.0, 80.0, 70.0])
    s = G.System(n, n_slots=1)
    s.synth_uniform(0, box, SEED)
    s.group_create_from_ranges("S", [(0, 9999)])
    pos = s.get_positions()
    got = s.group_all_distances("S", "S")
    assert got.shape == (10000, 10000) and np.all(np.diag(got) == 0.0)
    assert np.abs(got - got.T).max() <= 2e-6
    want = O.group_all_distances(pos, np.arange(10000), np.arange(10000), "xyz", box)
    assert np.abs(got - want).max() <= 1e-5   # magnitude from |d|^2 + gain of the best image: f32 cancellation ~7e-6 at 13 nm
    rng = np.random.default_rng(0)
    for _ in range(20):
        i, j = rng.integers(0, 10000, 2)
        assert abs(got[i, j] - brute_dist(pos[i], pos[j], box)) <= 2e-5
    # half the shortest lattice vector bounds every minimum-image distance from above only loosely; the long
    # diagonal bounds it strictly
    assert got.max() <= 0.5 * np.linalg.norm(lattice(box).sum(0)) + 1e-3
    s.close()
