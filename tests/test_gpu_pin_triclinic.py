"""Pins the TRICLINIC path of the library to the reference's own ORTHORHOMBIC arithmetic, and puts the benchmark's
full-size workload under the oracle.

The reference rejects non-orthogonal boxes (simbox.rs:230-236), so no reference value exists for them.  What does exist is
an equivalence: the same atoms, known as UNWRAPPED coordinates, are given

  * to the pinned orthorhombic path of the oracle (strict-orthogonal mode on: anything else raises) in a large
    orthorhombic box, where the compact groups are whole or broken across orthorhombic faces only -- rmsd.rs:425-603,
    iterators.rs:1152-1191,1404-1438, pinned by the reference's known answers in tests/test_oracle_golden.py;
  * to the HIP path, wrapped into a triclinic / truncated-octahedral / rhombic-dodecahedral cell: PBC-broken, every frame
    rigidly moved by a random rotation + a translation of several cells, and presented in three different periodic images
    (brick cell, parallelepiped cell, atoms scattered over neighbouring cells).

Both sides shift the group's centre of mass to their box centre, wrap, and subtract the box centre (rmsd.rs:430-445,479-492):
both end up with coordinates relative to the COM of the made-whole group.  Hence RMSD, rotation, COM (modulo a lattice vector)
and the fitted coordinates of the group (modulo the same lattice vector) must agree -- to 1e-5 nm / 5e-5 nm -- iff the
library's triclinic wrap, fractional Bai-Breen centre, box centre, minimum image and single-pass image proof implement what
the orthorhombic reference arithmetic means.  The library's own oracle definition of the triclinic extension is NOT consulted
in this file (test_gpu_triclinic_fullsize.py does that)."""
import zlib

import numpy as np
import pytest

import oracle_lib as O
from groan_rs_amd import workload as W

pytestmark = pytest.mark.gpu
SEED = 20260424

CELLS = {
    "triclinic_75_80_70": ([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]),
    "dodecahedron": ([6.0, 6.0, 6.0], [60.0, 60.0, 90.0]),
    "octahedron": ([6.0, 6.0, 6.0], [70.53, 109.47, 70.53]),
    "skewed_negative": ([6.5, 7.5, 6.0], [100.0, 95.0, 110.0]),
    "flat_60_70_80": ([8.0, 7.0, 3.0], [60.0, 70.0, 80.0]),          # cz = 2.48 nm: images two steps of c away can win (14 table entries)
}


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def lattice(box9):
    b = np.asarray(box9, np.float64)
    return np.array([[b[0], 0, 0], [b[5], b[1], 0], [b[7], b[8], b[2]]])    # rows = box vectors a, b, c


def rand_rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def image_of(x64, box9, kind, rng):
    """a periodic image of the unwrapped fp64 coordinates, computed here in fp64 (independent of the code under test)"""
    L = lattice(box9)
    if kind == "brick":        # GROMACS put_atoms_in_box: 0 <= z < cz, then 0 <= y < by, then 0 <= x < ax
        x = x64.copy()
        for d in (2, 1, 0):
            k = np.floor(x[:, d] / L[d, d])
            x -= k[:, None] * L[d][None, :]
        return x.astype(np.float32)
    frac = x64 @ np.linalg.inv(L)
    if kind == "parallelepiped":
        frac -= np.floor(frac)
    else:                      # "scattered": every atom in a random neighbouring cell
        frac = frac - np.floor(frac) + rng.integers(-2, 3, frac.shape)
    return (frac @ L).astype(np.float32)


def lattice_residual(d, box9):
    """distance of the vector d from the nearest lattice vector"""
    L = lattice(box9)
    f = np.asarray(d, np.float64) @ np.linalg.inv(L)
    return np.abs((f - np.rint(f)) @ L).max()


def blob(rng, n, radius):
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
    return v * (radius * rng.uniform(0, 1, (n, 1)) ** (1 / 3))


@pytest.mark.parametrize("cell", sorted(CELLS))
@pytest.mark.parametrize("n_atoms,group", [(900, "prefix"), (6000, "scattered"), (60000, "prefix")])
def test_triclinic_path_equals_pinned_orthorhombic_arithmetic(G, cell, n_atoms, group):
    lengths, angles = CELLS[cell]
    box_t = W.box_from_lengths_angles(lengths, angles)
    rng = np.random.default_rng(zlib.crc32(("%s/%d/%s" % (cell, n_atoms, group)).encode()))
    radius = 0.2 * float(min(box_t[0], box_t[1], box_t[2]))
    big = np.float32(4.0 * max(lengths))
    box_o = np.array([big, big, big, 0, 0, 0, 0, 0, 0], np.float32)
    # the group: a prefix (tile path, one-pass centre for >= 4096 atoms) or every third atom + a few ranges (gather path)
    if group == "prefix":
        idx = np.arange(int(0.8 * n_atoms))
    else:
        idx = np.unique(np.concatenate([np.arange(0, n_atoms, 3), np.arange(100, 400), np.arange(n_atoms - 50, n_atoms)]))
    S = idx.size
    masses = rng.choice(np.array([1.008, 12.011, 14.007, 15.999, 32.06], np.float32), n_atoms)
    # unwrapped coordinates: group = compact blob; the other atoms anywhere (they only ride along in the fit)
    u0 = rng.uniform(-3 * max(lengths), 3 * max(lengths), (n_atoms, 3))
    u0[idx] = blob(rng, S, radius) + rng.uniform(-10, 10, 3)
    nf = 3
    frames_u = []
    for f in range(nf):
        R, t = rand_rot(rng), rng.uniform(-3 * max(lengths), 3 * max(lengths), 3)       # several cells away
        c = u0[idx].mean(0)
        frames_u.append((u0 - c) @ R.T + c + t + rng.normal(0, 0.03, u0.shape))
    # ---- reference side: pinned orthorhombic oracle on the same atoms in a large orthorhombic box (wrapped into it, so the
    # group is broken across ORTHORHOMBIC faces whenever it straddles one)
    big_shift = rng.uniform(0, float(big), 3)
    ref_o = O.wrap_atoms((u0 + big_shift).astype(np.float32), np.arange(n_atoms), box_o)
    O.set_strict_orthogonal(True)
    # per-atom arithmetic in f32 exactly as pinned, sums in double: the reference's sequential f32 sums of coordinates of ~28 nm
    # lose 1e-4 nm by themselves already at 700 atoms (each add rounds at the magnitude of the running sum) -- that error is
    # the reference's, not a property of either path under comparison
    O.set_accumulate_f64(True)
    try:
        want = []
        for f in range(nf):
            cur_o = O.wrap_atoms((frames_u[f] + rng.uniform(0, float(big), 3)).astype(np.float32), np.arange(n_atoms), box_o)
            r_o, R_o = O.calc_rmsd(ref_o, masses, idx, box_o, cur_o, masses, idx, box_o)
            r_o2, fit_o = O.calc_rmsd_and_fit(ref_o, masses, idx, box_o, cur_o, masses, idx, box_o)
            want.append((r_o, R_o, fit_o))
        com_ref_o = O.get_center(ref_o, idx, box_o, mass=masses)
        cen_ref_o = O.get_center(ref_o, idx, box_o)
    finally:
        O.set_strict_orthogonal(False); O.set_accumulate_f64(False)
    # ---- HIP side: the same atoms in the triclinic cell, three periodic images
    for kind in ("brick", "parallelepiped", "scattered"):
        ref_t = image_of(u0, box_t, kind, rng)
        ref = G.System(n_atoms, masses=masses, box=box_t, positions=ref_t)
        cur = G.System(n_atoms, masses=masses, n_slots=nf)
        for s in (ref, cur):
            s.group_create_from_indices("G", idx.tolist())
        for f in range(nf):
            cur.set_frame(image_of(frames_u[f], box_t, kind, rng), box_t, slot=f)
        # centre of mass / of geometry of the reference: equal to the orthorhombic side's up to the (unknown) shift between the
        # two coordinate systems -- so compare DIFFERENCES: (com - centre) is frame-independent, and com itself mod lattice
        com_t, cen_t = ref.group_get_com("G"), ref.group_get_center("G")
        true_com = (u0[idx] * masses[idx, None].astype(np.float64)).sum(0) / masses[idx].astype(np.float64).sum()
        assert lattice_residual(com_t.astype(np.float64) - true_com, box_t) <= 1e-5, (cell, kind)
        assert lattice_residual(cen_t.astype(np.float64) - u0[idx].mean(0), box_t) <= 1e-5
        np.testing.assert_allclose(com_t.astype(np.float64) - cen_t, com_ref_o.astype(np.float64) - cen_ref_o, atol=1.5e-5, rtol=0)
        for exact in (False, True):
            plan = G.RMSDPlan(ref, cur, "G")
            plan.force_exact(exact)
            r, st, Rm = plan.rmsd(0, nf, return_rotation=True)
            assert (st == 0).all()
            if not exact:
                assert plan.last_fallbacks() == 0            # compact group: the single-pass image proof must hold
            for f in range(nf):
                assert abs(float(r[f]) - want[f][0]) <= 1e-5, (cell, kind, exact, f, float(r[f]), want[f][0])
                np.testing.assert_allclose(Rm[f], want[f][1], atol=2e-5, rtol=0)
            # fit: group atoms land on the reference group up to the lattice vector between the two sides' reference COMs
            before = [cur.get_positions(f) for f in range(nf)]
            rf, st = plan.rmsd_fit(0, nf)
            assert (st == 0).all()
            for f in range(nf):
                assert abs(float(rf[f]) - want[f][0]) <= 1e-5
                d = cur.get_positions(f)[idx].astype(np.float64) - want[f][2][idx].astype(np.float64)
                off = com_t.astype(np.float64) - com_ref_o.astype(np.float64)       # same for every atom and frame
                assert np.abs(d - off).max() <= 5e-5, (cell, kind, exact, f, np.abs(d - off).max())
                cur.set_frame(before[f], box_t, slot=f)      # un-fit for the next variant
            plan.close()
        ref.close(); cur.close()


@pytest.mark.parametrize("cell", sorted(CELLS))
def test_wrap_and_translate_land_in_the_brick_cell_on_the_same_lattice_site(G, cell):
    """atoms_wrap / atoms_translate in a triclinic cell, against the definition itself (no oracle): every atom ends inside
    0 <= x <= ax, 0 <= y <= by, 0 <= z <= cz (the closed upper end is the reference's, vector3d.rs:398-417) and differs
    from its input (+ the translation) by a lattice vector"""
    lengths, angles = CELLS[cell]
    box = W.box_from_lengths_angles(lengths, angles)
    rng = np.random.default_rng(7)
    n = 20000
    x = rng.uniform(-4 * max(lengths), 4 * max(lengths), (n, 3)).astype(np.float32)
    s = G.System(n, box=box, positions=x)
    s.atoms_wrap()
    w = s.get_positions()
    assert (w[:, 0] >= 0).all() and (w[:, 0] <= box[0]).all() and (w[:, 1] >= 0).all() and (w[:, 1] <= box[1]).all()
    assert (w[:, 2] >= 0).all() and (w[:, 2] <= box[2]).all()
    L = lattice(box)
    f = (w.astype(np.float64) - x.astype(np.float64)) @ np.linalg.inv(L)
    assert np.abs((f - np.rint(f)) @ L).max() <= 1e-5           # f32 rounding of ~30 nm coordinates: 2e-6 per operation
    v = np.array([3.3, -17.1, 8.25], np.float32)
    s.set_frame(x, box)
    s.atoms_translate(v)
    t = s.get_positions()
    f = (t.astype(np.float64) - x.astype(np.float64) - v.astype(np.float64)) @ np.linalg.inv(L)
    assert np.abs((f - np.rint(f)) @ L).max() <= 1e-5
    assert (t[:, 2] >= 0).all() and (t[:, 2] <= box[2]).all() and (t[:, 1] >= 0).all() and (t[:, 1] <= box[1]).all() and (t[:, 0] >= 0).all() and (t[:, 0] <= box[0]).all()
    s.close()


def test_config4_full_size_against_the_oracle(G):
    """BASELINE configs[3] exactly as bench.py runs it (1e6 atoms, rhombic dodecahedron d = 24.18 nm, blob of 0.2 x the
    shortest height, noise 0.05 nm, all atoms selected): two frames through gr_rmsd_fit_batch vs the oracle (sums in double:
    the reference's sequential f32 sums are off by ~1e-2 nm over 1e6 terms)"""
    n, nf = 1_000_000, 2
    box = W.c4_box()
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    frames = [cur.get_positions(f) for f in range(nf)]
    plan = G.RMSDPlan(ref, cur, "all")
    r, st = plan.rmsd_fit(0, nf)
    assert (st == 0).all() and plan.last_fallbacks() == 0
    idx = np.arange(n)
    with O.acc64():
        for f in range(nf):
            ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box)
            assert abs(float(r[f]) - ro) <= 1e-5, (f, float(r[f]), ro)
            assert np.abs(cur.get_positions(f) - want).max() <= 5e-5
    plan.close(); ref.close(); cur.close()


def test_config3_first_rows_against_brute_force_images(G):
    """BASELINE configs[2] box (24 x 23 x 22 nm, 75/80/70): rows of the 1e4 x 1e4 matrix against an fp64 search over
    9 x 9 x 9 lattice images -- the true minimum image, independent of the oracle's triclinic definition"""
    n = 1_000_000
    box = W.c3_box()
    s = G.System(n, n_slots=1)
    s.synth_uniform(0, box, W.SEED)
    s.group_create_from_ranges("S", [(0, 9999)])
    pos = s.get_positions()[:10000].astype(np.float64)
    got = s.group_all_distances("S", "S")
    L = lattice(box)
    ks = np.array([(i, j, k) for i in range(-4, 5) for j in range(-4, 5) for k in range(-4, 5)], np.float64) @ L
    rng = np.random.default_rng(3)
    for i in rng.integers(0, 10000, 6):
        d = pos[i][None, :] - pos                          # [1e4, 3]
        best = np.full(10000, np.inf)
        for c0 in range(0, ks.shape[0], 81):
            dd = d[:, None, :] + ks[None, c0:c0 + 81, :]
            best = np.minimum(best, (dd ** 2).sum(2).min(1))
        assert np.abs(got[i] - np.sqrt(best)).max() <= 2e-5
    s.close()


FLAT_CELLS = {
    # one box vector much shorter than the skew of the others: the closest image is 3-5 lattice steps from the brick-reduced
    # vector.  Rounds 1-3 built the image table from |i|, |j|, |k| <= 2 whatever the cell and returned longer vectors here.
    "flat_a": [12.8942, 29.4173, 3.27353, 0, 0, -4.28403, 0, -1.85917, -4.05997],
    "flat_b": [20.0702, 24.0458, 2.42171, 0, 0, -8.9196, 0, -6.97267, -6.8196],
    "flat_on_the_limits": [7.0228, 23.7485, 2.67607, 0, 0, 3.5114, 0, 3.5114, 11.8743],
    "mildly_flat": [12.0, 11.0, 5.0, 0, 0, 3.0, 0, -4.0, 4.5],
    # 14 / 16 entries that do NOT fit the (t, t + a) pair layout with its pads: the searches take every dot product (cand_pairs = 0)
    "unpaired_14": [14.4384, 13.3817, 2.28351, 0, 0, 1.70626, 0, 0.773115, 4.72202],
    "unpaired_16": [12.7645, 23.762, 4.77463, 0, 0, -4.08967, 0, 4.0417, -5.41221],
}


@pytest.mark.parametrize("cell", list(FLAT_CELLS))
def test_flat_cells_are_exact_or_refused(G, cell):
    """pair distances (plain and self-matrix kernels, XYZ and a 2-D dimension) and the centre of a group in flat cells: equal to an
    fp64 search over 15 x 15 x 15 lattice images -- or the cell is refused with "box too skewed for the minimum-image table"
    (more than 16 +- pairs of lattice vectors can win): never a silently longer vector"""
    import itertools
    box = np.array(FLAT_CELLS[cell], np.float32)
    L = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    rng = np.random.default_rng(4)
    n = 256
    pos = (rng.random((n, 3)) @ L).astype(np.float32)
    s = G.System(n, n_slots=1)
    s.set_frame(pos, box, slot=0)
    s.group_create_from_ranges("A", [(0, 99)])
    s.group_create_from_ranges("B", [(100, n - 1)])
    s.group_create_from_ranges("S", [(0, n - 1)])
    try:
        d_ab = s.group_all_distances("A", "B", G.Dimension.XYZ)
    except G.DeviceError as e:
        assert "skewed" in str(e), e
        assert cell not in ("mildly_flat", "unpaired_14", "unpaired_16")            # (<= 16 +- pairs: must be supported)
        s.close()
        return
    ks = np.array(list(itertools.product(range(-7, 8), repeat=3)), np.float64) @ L                # 3375 images
    def brute(a, b):
        d = pos[a].astype(np.float64)[:, None, :] - pos[b].astype(np.float64)[None, :, :]
        best = np.full(d.shape[:2], np.inf)
        vec = np.zeros(d.shape)
        for t in ks:
            v = d + t
            r = (v ** 2).sum(-1)
            m = r < best
            best[m] = r[m]; vec[m] = v[m]
        return np.sqrt(best), vec
    want, _ = brute(np.arange(100), np.arange(100, n))
    tol = 2e-6 + 2.5e-7 * (box[:3].astype(np.float64) ** 2).sum() / 4 / np.maximum(want, 1e-3)   # (the length-only search: DESIGN.md "Pair distances")
    assert (np.abs(d_ab - want) <= tol).all(), float(np.abs(d_ab - want).max())
    d_ss = s.group_all_distances("S", "S", G.Dimension.XYZ)
    want_ss, vec_ss = brute(np.arange(n), np.arange(n))
    tol = 2e-6 + 2.5e-7 * (box[:3].astype(np.float64) ** 2).sum() / 4 / np.maximum(want_ss, 1e-3)
    assert (np.abs(d_ss - want_ss) <= tol).all(), float(np.abs(d_ss - want_ss).max())
    d_xy = s.group_all_distances("S", "S", G.Dimension.XY)
    ok = np.abs(d_xy - np.hypot(vec_ss[..., 0], vec_ss[..., 1])) <= 5e-6
    assert ok.mean() > 0.999, ok.mean()          # (pairs whose two best images tie to rounding may pick either one)
    s.close()
