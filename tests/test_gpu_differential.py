"""Differential test of the GPU path against the oracle on ADVERSARIAL inputs: coordinates exactly on cell faces and half-box
separations, one ulp either side of them, tiny negatives, atoms several boxes away, one- to three-atom groups, coincident
atoms, very small and very flat boxes.  Every case is generated from a seed; the operations are the reference's
wrap / translate (bit-exact; vector3d.rs:380-417, iterators.rs:1520-1553), centres (iterators.rs:886-1438), distances
(analysis.rs:348-471, vector3d.rs:458-486) and calc_rmsd(_and_fit) (rmsd.rs:75-166)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
DIMS = ["X", "Y", "Z", "XY", "XZ", "YZ", "XYZ"]


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def nasty_positions(rng, n, L):
    """coordinates drawn per component from a mix of distributions that sit on the decision points of the closed forms"""
    pos = np.empty((n, 3), np.float32)
    for a in range(3):
        kind = rng.integers(0, 8, n)
        u = rng.random(n)
        base = (u * L[a]).astype(np.float32)                                   # 0: inside the cell
        v = base.copy()
        faces = np.float32(L[a]) * rng.choice(np.float32([0.0, 0.5, 1.0, -1.0, 2.0, -0.5, 1.5]), n)
        m = kind == 1; v[m] = faces[m]                                         # 1: exactly on a face / half box / next cell
        m = kind == 2; v[m] = np.nextafter(faces[m], np.float32(np.inf))       # 2, 3: one ulp either side
        m = kind == 3; v[m] = np.nextafter(faces[m], np.float32(-np.inf))
        m = kind == 4; v[m] = -np.float32(10.0) ** rng.integers(-12, -5, m.sum()).astype(np.float32)   # 4: tiny negatives
        m = kind == 5; v[m] = base[m] + np.float32(L[a]) * rng.integers(-4, 5, m.sum()).astype(np.float32)   # 5: a few cells away
        m = kind == 6; v[m] = base[m] * np.float32(1e-3)                       # 6: crowded near the origin
        pos[:, a] = v                                                          # 7: inside the cell again
    dup = rng.random(n) < 0.1                                                  # coincident atoms
    if n > 1:
        pos[dup] = pos[rng.integers(0, n, dup.sum())]
    return pos


def eq_bits(a, b):
    """bit equality up to the sign of a zero"""
    return np.array_equal(a.view(np.uint32) & 0x7fffffff, b.view(np.uint32) & 0x7fffffff)


@pytest.mark.parametrize("seed", range(24))
def test_orthorhombic_operations_on_nasty_inputs(G, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 257, 300]))
    L = np.float32(rng.choice([0.37, 1.0, 3.0, 6.5, 13.01327, 30.0], 3) * rng.choice([1.0, 1.0, 0.05, 1.7], 3))
    box = np.array([L[0], L[1], L[2], 0, 0, 0, 0, 0, 0], np.float32)
    pos = nasty_positions(rng, n, L)
    masses = rng.choice(np.float32([1.008, 12.011, 15.999, 72.0, 0.0005]), n).astype(np.float32)
    s = G.System(n, masses=masses, n_slots=2)
    a0 = int(rng.integers(0, n)); a1 = int(rng.integers(a0, n))
    b0 = int(rng.integers(0, n)); b1 = int(rng.integers(b0, n))
    s.group_create_from_ranges("A", [(a0, a1)]); s.group_create_from_ranges("B", [(b0, b1)])
    ia, ib, iall = np.arange(a0, a1 + 1), np.arange(b0, b1 + 1), np.arange(n)
    # ---- wrap / translate: bit-exact
    s.set_frame(pos, box)
    s.atoms_wrap()
    assert eq_bits(s.get_positions(), O.wrap_atoms(pos, iall, box)), "atoms_wrap"
    v = (rng.normal(0, 1, 3) * L * rng.choice([1e-6, 0.3, 2.5])).astype(np.float32)
    s.set_frame(pos, box)
    s.group_translate("A", v)
    assert eq_bits(s.get_positions(), O.translate(pos, ia, v, box)), "group_translate"
    # ---- centres
    s.set_frame(pos, box)
    with O.acc64():
        for weighted in (False, True):
            m = masses if weighted else None
            got = np.array(s.group_get_com_naive("A") if weighted else s.group_get_center_naive("A"))
            want = O.center_naive(pos, ia, mass=m)
            np.testing.assert_allclose(got, want, atol=TOL, rtol=2e-7, err_msg="naive")
            if ia.size <= 64:   # (the circular mean of many scattered atoms is ill-conditioned: a short resultant amplifies the last bit of every angle)
                got = np.array(s.group_estimate_com("A") if weighted else s.group_estimate_center("A"))
                want = O.estimate_center(pos, ia, box, mass=m)
                d = np.abs(got - want); d = np.minimum(d, np.abs(d - L))       # (0 and L are the same point of the circle)
                assert d.max() <= 2e-5 * max(1.0, float(L.max())), ("estimate", got, want)
    # ---- distances: every dimension, the whole matrix
    for dim in DIMS:
        got = s.group_all_distances("A", "B", G.Dimension[dim])
        want = O.group_all_distances(pos, ia, ib, dim.lower(), box)
        if len(dim) == 1:
            assert eq_bits(got, want), dim          # signed 1-D distances: the min_image loops themselves, bit for bit
        else:
            np.testing.assert_allclose(got, want, atol=3e-6 * max(1.0, float(L.max())), rtol=2e-6, err_msg=dim)
    i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
    assert abs(s.atoms_distance(i, j) - O.distance(pos[i], pos[j], "xyz", box)) <= 3e-6 * max(1.0, float(L.max()))
    s.close()


@pytest.mark.parametrize("seed", range(12))
def test_rmsd_of_compact_groups_anywhere_in_the_cell(G, seed):
    """calc_rmsd / calc_rmsd_and_fit for a compact group broken over the periodic boundary in both systems, random rotation,
    small boxes (the group fills up to 40 % of the cell), atoms of the rest of the system anywhere"""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([12, 61, 200, 1500]))
    ns = max(4, n // 2)
    L = np.float32(rng.choice([3.0, 6.5, 13.0], 3))
    box = np.array([L[0], L[1], L[2], 0, 0, 0, 0, 0, 0], np.float32)
    core = rng.normal(0, 0.07 * float(L.min()), (ns, 3))
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    masses = rng.choice(np.float32([1.008, 12.011, 15.999, 32.06]), n).astype(np.float32)
    def frame(rot, noise):
        p = np.empty((n, 3), np.float64)
        p[:ns] = core @ rot.T + rng.random(3) * L + rng.normal(0, noise, (ns, 3))
        p[ns:] = rng.random((n - ns, 3)) * L * 3 - L
        out = p.astype(np.float32)
        out[:ns] = O.wrap_atoms(out[:ns].copy(), np.arange(ns), box)
        return out
    ref_pos, cur_pos = frame(np.eye(3), 0.0), frame(q, 0.02)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    cur = G.System(n, masses=masses, box=box, positions=cur_pos)
    for sy in (ref, cur):
        sy.group_create_from_ranges("G", [(0, ns - 1)])
    idx = np.arange(ns)
    with O.acc64():
        want_r, want_fit = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, cur_pos, masses, idx, box)
    got = cur.calc_rmsd(ref, "G")
    assert abs(got - want_r) <= TOL, (got, want_r)
    got2 = cur.calc_rmsd_and_fit(ref, "G")
    assert abs(got2 - want_r) <= TOL
    np.testing.assert_allclose(cur.get_positions(), want_fit, atol=5e-5, rtol=0)
    ref.close(); cur.close()


@pytest.mark.parametrize("seed", range(10))
def test_geometry_selection_and_cutoff_pairs_with_atoms_on_the_boundaries(G, seed):
    """Shape::inside (src/structures/shape.rs:110-505) and the cell-grid pair search (cellgrid.rs:301-409) decide membership by
    comparing a distance with a threshold: atoms are placed EXACTLY on sphere / cylinder radii, box faces of the rectangular
    shape, prism faces and at the cut-off distance (and one ulp either side); the index lists must be identical"""
    rng = np.random.default_rng(9000 + seed)
    n = 400
    L = np.float32(rng.choice([4.0, 6.5, 9.0], 3))
    box = np.array([L[0], L[1], L[2], 0, 0, 0, 0, 0, 0], np.float32)
    pos = (rng.random((n, 3)) * L).astype(np.float32)
    centre = (rng.random(3) * L).astype(np.float32)
    radius = np.float32(rng.choice([0.75, 1.3, 2.0]))
    for k in range(0, 120):                                       # on / next to the sphere and cylinder radius, through the boundary
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        r = radius if k % 3 == 0 else np.nextafter(radius, np.float32(np.inf if k % 3 == 1 else -np.inf))
        if k % 2:
            d[1] = 0.0; d /= np.linalg.norm(d)                     # in the plane of a Y cylinder
        pos[k] = centre + np.float32(r) * d.astype(np.float32)
    size = np.float32([1.5, 2.25, 1.0])
    for k in range(120, 180):                                     # faces of the rectangular shape (origin corner + size)
        a = k % 3
        p = centre + (rng.random(3) * size).astype(np.float32)
        p[a] = centre[a] + (size[a] if k % 2 else np.float32(0.0))
        if k % 5 == 0:
            p[a] = np.nextafter(p[a], np.float32(np.inf))
        pos[k] = p
    b1, b2, b3 = centre, centre + np.float32([2.0, 0.5, 0.0]), centre + np.float32([0.5, 2.0, 0.0])
    for k in range(180, 240):                                     # on the slanted faces of the prism (sign of a cross product ~ 0)
        e0, e1 = ((b1, b2), (b2, b3), (b3, b1))[k % 3]
        p = (e0 + np.float32(rng.random()) * (e1 - e0)).astype(np.float32)
        p[2] = centre[2] + np.float32(rng.random() * 1.5)
        pos[k] = p
    pos = O.wrap_atoms(pos, np.arange(n), box)
    s = G.System(n, n_slots=1)
    s.set_frame(pos, box)
    s.group_create_from_ranges("src", [(0, n - 1)])
    idx = np.arange(n)
    specs = [{"kind": "sphere", "position": centre.tolist(), "radius": float(radius)},
             {"kind": "cylinder", "position": centre.tolist(), "radius": float(radius), "height": 2.5, "orientation": "Y"},
             {"kind": "rectangular", "position": centre.tolist(), "size": size.tolist()},
             {"kind": "prism", "base1": centre.tolist(), "base2": (centre + np.float32([2.0, 0.5, 0.0])).tolist(), "base3": (centre + np.float32([0.5, 2.0, 0.0])).tolist(), "height": 1.5}]
    def build(sp):
        k = sp["kind"]
        if k == "sphere": return G.Sphere(sp["position"], sp["radius"])
        if k == "rectangular": return G.Rectangular(sp["position"], *sp["size"])
        if k == "cylinder": return G.Cylinder(sp["position"], sp["radius"], sp["height"], G.Dimension[sp["orientation"]])
        return G.TriangularPrism(sp["base1"], sp["base2"], sp["base3"], sp["height"])
    for sp in specs:
        for naive in ((False,) if sp["kind"] == "prism" else (False, True)):    # (the reference has no NaiveShape for the prism)
            s.group_create_from_geometries("sel", "src", [build(sp)], naive=naive)
            got = np.array(list(s.group_container("sel")), np.uint64)
            want = O.group_from_geometries(pos, idx, box, [sp], naive=naive)
            assert np.array_equal(got, want), (sp["kind"], naive, np.setxor1d(got, want))
    # pairs at exactly the cut-off: atom k + 200 sits at distance `cut` (or an ulp off) from atom k
    cut = np.float32(0.4)
    for k in range(40):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        r = cut if k % 3 == 0 else np.nextafter(cut, np.float32(np.inf if k % 3 == 1 else -np.inf))
        pos[200 + k] = pos[k] + np.float32(r) * d.astype(np.float32)
    pos = O.wrap_atoms(pos, np.arange(n), box)
    s.set_frame(pos, box)
    s.group_create_from_ranges("g1", [(0, 199)]); s.group_create_from_ranges("g2", [(150, n - 1)])
    gi, gj, gd = s.group_pairs_within("g1", "g2", float(cut))
    wi, wj, wd = O.pairs_within(pos, np.arange(0, 200), np.arange(150, n), box, float(cut))
    assert np.array_equal(gi.astype(np.uint64), wi) and np.array_equal(gj.astype(np.uint64), wj), (gi.size, wi.size)
    s.close()


@pytest.mark.parametrize("seed", range(8))
def test_triclinic_operations_on_nasty_inputs(G, seed):
    """the triclinic extension (parity unpinned: the reference refuses such boxes) against the oracle's restatement of the
    same definitions: wrapped positions must be the same point of the lattice, centres and distances agree to 1e-5 nm"""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([3, 17, 64, 200]))
    angles = [[60.0, 60.0, 90.0], [70.53, 109.47, 70.53], [75.0, 80.0, 70.0], [90.0, 90.0, 60.0]][seed % 4]
    l0 = float(rng.choice([3.0, 6.5, 12.0]))
    box = O.box_from_lengths_angles([l0, l0, l0] if seed % 4 < 2 else [l0, 0.9 * l0, 0.8 * l0], angles)
    boxm = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    frac = rng.random((n, 3))
    kind = rng.integers(0, 5, (n, 3))
    frac = np.where(kind == 1, rng.choice([0.0, 0.5, 1.0], (n, 3)), frac)          # on faces / half way
    frac = np.where(kind == 2, frac + rng.integers(-3, 4, (n, 3)), frac)           # a few cells away
    frac = np.where(kind == 3, -10.0 ** rng.integers(-9, -5, (n, 3)), frac)        # tiny negatives
    pos = (frac @ boxm).astype(np.float32)
    masses = rng.choice(np.float32([1.008, 12.011, 15.999]), n).astype(np.float32)
    s = G.System(n, masses=masses, n_slots=1)
    s.group_create_from_ranges("A", [(0, n // 2)]); s.group_create_from_ranges("B", [(n // 3, n - 1)])
    ia, ib, iall = np.arange(0, n // 2 + 1), np.arange(n // 3, n), np.arange(n)
    scale = max(1.0, l0)
    s.set_frame(pos, box)
    s.atoms_wrap()
    got, want = s.get_positions(), O.wrap_atoms(pos, iall, box)
    for i in range(n):                                                             # the same lattice point (a face atom may sit on either face)
        assert O.distance(got[i], want[i], "xyz", box) <= 2e-5 * scale, (i, got[i], want[i])
    s.set_frame(pos, box)
    for dim in ("XYZ", "X", "YZ"):
        g = s.group_all_distances("A", "B", G.Dimension[dim])
        w = O.group_all_distances(pos, ia, ib, dim.lower(), box)
        if dim == "XYZ":
            np.testing.assert_allclose(g, w, atol=1e-5 * scale, rtol=2e-6)
        else:   # components of the minimum image: where two images tie (half-box separations) either is a minimum image
            bad = np.abs(g - w) > 1e-5 * scale
            assert bad.mean() <= 0.2, (dim, bad.mean())
    with O.acc64():
        got = np.array(s.group_get_com_naive("A"))
        np.testing.assert_allclose(got, O.center_naive(pos, ia, mass=masses), atol=TOL * scale, rtol=2e-7)
    s.close()


@pytest.mark.parametrize("ns", [1, 2, 3, 4])
@pytest.mark.parametrize("flat", [False, True])
def test_rmsd_of_degenerate_groups(G, ns, flat):
    """one, two (collinear), three (planar) atoms and flat 4-atom groups: the covariance matrix is rank deficient, the
    optimal rotation is not unique for rank <= 1, but the RMSD (rmsd.rs:592-599) and the fitted positions of the group's own
    atoms are; for rank 2 the determinant correction (rmsd.rs:576-583) makes the whole rotation unique"""
    rng = np.random.default_rng(40 + ns + 10 * flat)
    n = 40
    box = np.array([5.0, 6.0, 7.0, 0, 0, 0, 0, 0, 0], np.float32)
    masses = rng.choice(np.float32([1.008, 12.011, 15.999]), n).astype(np.float32)
    for trial in range(6):
        core = rng.normal(0, 0.4, (ns, 3))
        if flat:
            core[:, 2] = 0.0
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        ref_pos = (rng.random((n, 3)) * box[:3]).astype(np.float32)
        cur_pos = (rng.random((n, 3)) * box[:3]).astype(np.float32)
        ref_pos[:ns] = (core + [2.5, 3.0, 3.5]).astype(np.float32)
        cur_pos[:ns] = O.wrap_atoms((core @ q.T + rng.random(3) * box[:3] + rng.normal(0, 0.03, (ns, 3))).astype(np.float32), np.arange(ns), box)
        ref = G.System(n, masses=masses, box=box, positions=ref_pos)
        cur = G.System(n, masses=masses, box=box, positions=cur_pos)
        for sy in (ref, cur):
            sy.group_create_from_ranges("G", [(0, ns - 1)])
        idx = np.arange(ns)
        with O.acc64():
            want_r, want_fit = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, cur_pos, masses, idx, box)
        assert abs(cur.calc_rmsd(ref, "G") - want_r) <= TOL, (ns, flat, trial)
        assert abs(cur.calc_rmsd_and_fit(ref, "G") - want_r) <= TOL
        got = cur.get_positions()
        np.testing.assert_allclose(got[:ns], want_fit[:ns], atol=5e-5, rtol=0)        # the group itself: unique
        rank = np.linalg.matrix_rank(core - core.mean(0), tol=1e-6) if ns > 1 else 0
        if rank >= 2:                                                                  # the whole rotation is unique: every atom
            np.testing.assert_allclose(got, want_fit, atol=2e-4, rtol=0)
        ref.close(); cur.close()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4099])
def test_sizes_around_the_tile_boundaries(G, n):
    """systems and selections whose sizes straddle the 4-atom lane groups, the 64-lane wavefront and the 256-atom tile: batched
    RMSD-fit (two-pass and closed-form paths), centres, wrap -- against the oracle, every atom of the ragged tail included"""
    rng = np.random.default_rng(600 + n)
    box = O.box_from_lengths_angles([6.0, 6.0, 6.0], [60.0, 60.0, 90.0]) if n % 2 else np.array([6.0, 5.5, 5.0, 0, 0, 0, 0, 0, 0], np.float32)
    boxm = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    masses = rng.choice(np.float32([1.008, 12.011, 15.999]), n).astype(np.float32)
    core = rng.normal(0, 0.35, (n, 3))
    nf = 3
    ref_pos = (core + np.array([0.5, 0.5, 0.5]) @ boxm).astype(np.float32)
    cur = G.System(n, masses=masses, n_slots=nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    sel = (0, n - 1) if n < 8 else (1, n - 2)                  # interior selection: the first and last lane groups are partial
    idx = np.arange(sel[0], sel[1] + 1)
    for sy in (ref, cur):
        sy.group_create_from_ranges("S", [sel])
    frames = []
    for f in range(nf):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        x = (core @ q.T + rng.random(3) @ boxm + rng.normal(0, 0.02, (n, 3))).astype(np.float32)
        x = O.wrap_atoms(x, np.arange(n), box)
        frames.append(x); cur.set_frame(x, box, slot=f)
    with O.acc64():
        want = [O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box) for f in range(nf)]
        coms = [O.get_center(frames[f], idx, box, mass=masses) for f in range(nf)]
    got_com, st = cur.group_get_com_batch("S", 0, nf)
    assert (st == 0).all()
    for f in range(nf):
        assert np.abs(got_com[f] - coms[f]).max() <= TOL
    plan = G.RMSDPlan(ref, cur, "S")
    r, st = plan.rmsd(0, nf)                                    # closed-form single pass
    assert (st == 0).all() and np.abs(r - [w[0] for w in want]).max() <= TOL
    r, st = plan.rmsd_fit(0, nf)                                # two-pass
    assert (st == 0).all() and np.abs(r - [w[0] for w in want]).max() <= TOL
    rank = np.linalg.matrix_rank(core[idx] - core[idx].mean(0), tol=1e-6) if idx.size > 1 else 0
    for f in range(nf):
        got = cur.get_positions(f)
        np.testing.assert_allclose(got[idx], want[f][1][idx], atol=5e-5, rtol=0)
        if rank >= 2:
            np.testing.assert_allclose(got, want[f][1], atol=2e-4, rtol=0)
    plan.close(); ref.close(); cur.close()


def test_results_are_bitwise_reproducible(G):
    """no atomics on floating-point data anywhere: partial sums are combined in a fixed order, so the same frames give the same
    bits on every run -- RMSD, rotation, fitted coordinates, centres (large enough for every multi-workgroup reduction)"""
    rng = np.random.default_rng(77)
    n, nf = 300_000, 12
    box = O.box_from_lengths_angles([12.0, 12.0, 12.0], [60.0, 60.0, 90.0])
    masses = rng.choice(np.float32([1.008, 12.011, 15.999]), n).astype(np.float32)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, 0.2 * float(min(box[:3])), 3)
    ref = None
    runs = []
    for rep in range(3):
        cur.synth_frames(nf, 0, nf, 0, 0.05, 3)
        if ref is None:
            ref = G.System(n, masses=masses, box=box, positions=cur.get_positions(nf))
        plan = G.RMSDPlan(ref, cur, "all")
        com, _ = cur.group_get_com_batch("all", 0, nf)
        r0, _ = plan.rmsd(0, nf)
        r1, _ = plan.rmsd_fit(0, nf)
        runs.append((com.copy(), r0.copy(), r1.copy(), [cur.get_positions(f) for f in (0, nf - 1)]))
        plan.close()
    for rep in (1, 2):
        assert np.array_equal(runs[0][0].view(np.uint32), runs[rep][0].view(np.uint32))
        assert np.array_equal(runs[0][1].view(np.uint32), runs[rep][1].view(np.uint32))
        assert np.array_equal(runs[0][2].view(np.uint32), runs[rep][2].view(np.uint32))
        for a, b in zip(runs[0][3], runs[rep][3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ref.close(); cur.close()


@pytest.mark.parametrize("seed", range(10))
def test_centring_and_group_distance_on_nasty_inputs(G, seed):
    """atoms_center / atoms_center_mass in every Dimension (utility.rs:109-185: Bai-Breen estimate of the reference group, masked
    shift, translate + wrap of every atom) and group_distance (analysis.rs:348-360: distance between two get_center results) on
    the adversarial coordinate mix; the reference group is compact (a scattered group's circular mean is ill-conditioned)"""
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.choice([8, 40, 150, 300]))
    L = np.float32(rng.choice([1.0, 3.0, 6.5, 13.0], 3))
    box = np.array([L[0], L[1], L[2], 0, 0, 0, 0, 0, 0], np.float32)
    pos = nasty_positions(rng, n, L)
    nref = max(3, n // 4)
    centre = (rng.random(3) * L).astype(np.float32)
    pos[:nref] = O.wrap_atoms((centre + rng.normal(0, 0.05 * float(L.min()), (nref, 3))).astype(np.float32), np.arange(nref), box)
    pos[nref:2 * nref] = O.wrap_atoms((centre * np.float32(0.37) + rng.normal(0, 0.04 * float(L.min()), (min(nref, n - nref), 3))[: max(0, min(nref, n - nref))]).astype(np.float32), np.arange(max(0, min(nref, n - nref))), box) if n >= 2 * nref else pos[nref:2 * nref]
    masses = rng.choice(np.float32([1.008, 12.011, 15.999, 72.0]), n).astype(np.float32)
    s = G.System(n, masses=masses, n_slots=1)
    s.group_create_from_ranges("R", [(0, nref - 1)])
    if n >= 2 * nref:
        s.group_create_from_ranges("Q", [(nref, 2 * nref - 1)])
    iref = np.arange(nref)
    tol = 3e-5 * max(1.0, float(L.max()))
    for dim in DIMS:
        for weighted in (False, True):
            s.set_frame(pos, box)
            (s.atoms_center_mass if weighted else s.atoms_center)("R", G.Dimension[dim])
            got = s.get_positions()
            with O.acc64():
                want = O.atoms_center(pos, iref, dim.lower(), box, mass=masses if weighted else None)
            d = np.abs(got - want); d = np.minimum(d, np.abs(d - L))        # an atom may land on either face of the cell
            assert d.max() <= tol, (dim, weighted, float(d.max()))
    if n >= 2 * nref:
        s.set_frame(pos, box)
        iq = np.arange(nref, 2 * nref)
        with O.acc64():
            c1, c2 = O.get_center(pos, iref, box), O.get_center(pos, iq, box)
        for dim in DIMS:
            want = O.distance(c1, c2, dim.lower(), box)
            got = s.group_distance("R", "Q", G.Dimension[dim])
            assert min(abs(got - want), abs(abs(got - want) - float(L[{"X": 0, "Y": 1, "Z": 2}.get(dim, 0)]))) <= tol, (dim, got, want)
    s.close()
