"""Pin the CPU oracle (oracle/groan_oracle.c) against the reference's own known-answer tests.

Every expected value below is copied from an assertion in the reference's in-file unit tests
(file:line given, relative to the reference root) and is compared with the margin the reference
itself uses (float_cmp default = f32::EPSILON or 4 ulps unless an epsilon is stated).
Inputs are the committed fixtures under tests/golden/ (decoded from the reference's data files by
tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest

import oracle_lib as O
from conftest import assert_approx, approx_f32

BOX10 = [10.0, 10.0, 10.0]
BOX4 = [4.0, 4.0, 4.0]


# ----------------------------------------------------------------------------- vector3d.rs
def test_wrap_vector3d_rs_1017_1038():
    box = [2.0, 2.0, 2.0]
    w = O.wrap([-1.0, 1.5, 3.0], box)
    assert_approx(w[0], 1.0); assert_approx(w[1], 1.5); assert_approx(w[2], 1.0)
    w = O.wrap([2.0, 2.2, -0.3], box)
    assert_approx(w[0], 2.0); assert_approx(w[1], 0.2); assert_approx(w[2], 1.7)
    w = O.wrap([-54.2, 77.8, 124.5], box)
    assert_approx(w[0], 1.8, 1e-5); assert_approx(w[1], 1.8, 1e-5); assert_approx(w[2], 0.5, 1e-5)
    # doctest vector3d.rs:371-377
    w = O.wrap([-0.5, 2.0, 4.2], [4.0, 4.0, 4.0])
    assert_approx(w[0], 3.5, 1e-5); assert_approx(w[1], 2.0, 1e-5); assert_approx(w[2], 0.2, 1e-5)


@pytest.mark.parametrize("dim,exp12,exp21", [
    ("x", 1.5, -1.5), ("y", -0.2, 0.2), ("z", -1.8, 1.8), ("xy", 1.51327, 1.51327),
    ("xz", 2.34307, 2.34307), ("yz", 1.81108, 1.81108), ("xyz", 2.351595, 2.351595), ("none", 0.0, 0.0),
])
def test_distance_vector3d_rs_1040_1207(dim, exp12, exp21):
    p1, p2 = [1.0, 3.9, 2.6], [3.5, 0.1, 0.4]
    assert_approx(O.distance(p1, p2, dim, BOX4), exp12, 1e-5)
    assert_approx(O.distance(p2, p1, dim, BOX4), exp21, 1e-5)


def test_distance_outofbox_vector3d_rs_1186_1196():
    p1, p2 = [-1.0, 4.5, 2.3], [3.5, -0.5, 4.2]
    assert_approx(O.distance(p1, p2, "x", BOX4), -0.5)
    assert_approx(O.distance(p1, p2, "y", BOX4), 1.0)
    assert_approx(O.distance(p1, p2, "z", BOX4), -1.9)


def test_distance_doctests_vector3d_rs_433_455():
    assert_approx(O.distance([1.0, 2.0, 3.0], [3.5, 1.0, 2.0], "x", BOX4), 1.5)
    assert_approx(O.distance([3.5, 1.0, 2.0], [1.0, 2.0, 3.0], "x", BOX4), -1.5)
    assert_approx(O.distance_naive([1.0, 2.0, 3.0], [3.5, 1.0, 2.0], "xy"), 2.692582)
    assert_approx(O.distance_naive([1.0, 2.0, 3.0], [3.5, 1.0, 2.0], "x"), -2.5)


@pytest.mark.parametrize("p1,p2,exp", [
    ([4.0, 4.0, 5.0], [5.0, 5.0, 3.0], [1.0, 1.0, -2.0]),      # :1361-1372
    ([3.0, 0.0, 7.0], [1.0, 2.0, 1.0], [-2.0, 2.0, 4.0]),      # :1374-1385
    ([1.0, 2.0, 5.0], [9.0, 8.0, 6.0], [-2.0, -4.0, 1.0]),     # :1387-1398
    ([8.0, 9.0, 2.0], [1.0, 3.0, 9.0], [3.0, 4.0, -3.0]),      # :1400-1411
    ([0.0, 3.0, 10.0], [10.0, 3.0, 0.0], [0.0, 0.0, 0.0]),     # :1413-1424
])
def test_vector_to_vector3d_rs_1361_1424(p1, p2, exp):
    v = O.vector_to(p1, p2, BOX10)
    for a, b in zip(v, exp):
        assert_approx(a, b)


def test_vector_to_equidistant_vector3d_rs_1426_1437():
    v = O.vector_to([7.0, 4.0, 3.0], [2.0, 5.0, 2.0], BOX10)
    assert_approx(abs(v[0]), 5.0); assert_approx(v[1], 1.0); assert_approx(v[2], -1.0)
    v = O.vector_to([1.0, 2.0, 3.0], [3.0, 2.0, 1.0], [3.5, 5.0, 5.0])  # doctest :550-558
    assert_approx(v[0], -1.5); assert_approx(v[1], 0.0); assert_approx(v[2], -2.0)


# ----------------------------------------------------------------------------- simbox.rs
def test_box_from_lengths_angles_simbox_rs_81_95():
    b = O.box_from_lengths_angles([5.0, 4.0, 3.0], [80.0, 70.0, 120.0])
    exp = [5.0, 3.464102, 2.553768, 0.0, 0.0, -2.0, 0.0, 1.026060, 1.193930]
    for a, e in zip(b, exp):
        assert_approx(a, e, 1e-4)


# ----------------------------------------------------------------------------- container.rs
def _b(x):
    return [tuple(int(v) for v in r) for r in np.asarray(x).tolist()]


def test_container_from_indices_container_rs_517_573():
    assert _b(O.container_from_indices([6, 2, 13, 1, 10, 8, 3, 12, 7, 14, 15], 16)) == [(1, 3), (6, 8), (10, 10), (12, 15)]
    assert _b(O.container_from_indices([], 16)) == []
    dup = [1, 6, 3, 2, 13, 1, 10, 8, 3, 12, 7, 14, 15, 10]
    assert _b(O.container_from_indices(dup, 16)) == [(1, 3), (6, 8), (10, 10), (12, 15)]
    assert _b(O.container_from_indices(dup, 15)) == [(1, 3), (6, 8), (10, 10), (12, 14)]
    cx = [11, 1, 2, 3, 20, 5, 0, 5, 4, 18, 6, 19, 1, 13, 20, 27]
    assert _b(O.container_from_indices(cx, 20)) == [(0, 6), (11, 11), (13, 13), (18, 19)]


@pytest.mark.parametrize("ranges,n,exp", [
    ([(20, 32)], 33, [(20, 32)]),                                             # :598-604
    ([], 33, []),                                                             # :607-612
    ([(20, 32), (64, 64), (84, 143)], 1028, [(20, 32), (64, 64), (84, 143)]),  # :615-623
    ([(20, 32), (33, 42)], 1028, [(20, 42)]),                                 # :626-632
    ([(20, 32), (24, 42)], 1028, [(20, 42)]),
    ([(20, 35), (20, 32)], 1028, [(20, 35)]),
    ([(20, 32), (28, 30)], 1028, [(20, 32)]),
    ([(28, 30), (20, 32)], 1028, [(20, 32)]),
    ([(20, 32), (32, 42)], 1028, [(20, 42)]),
    ([(64, 128), (5, 32), (1, 25), (129, 133), (133, 200), (35, 78), (10, 15), (1033, 1055)], 1028, [(1, 32), (35, 200)]),
    ([(1, 25), (0, 1), (0, 0), (0, 34)], 1028, [(0, 34)]),
    ([(32, 25), (14, 17)], 1028, [(14, 17)]),
    ([(543, 1020), (1000, 1432)], 1028, [(543, 1027)]),
    ([(543, 1020), (1043, 1432)], 1028, [(543, 1020)]),
    ([(0, 43), (1006, 1432)], 1028, [(0, 43), (1006, 1027)]),
    ([(5, 5), (4, 4), (11, 11), (12, 12), (0, 0), (1, 1), (6, 6), (2, 2), (3, 3), (7, 7), (13, 13), (10, 10), (8, 8), (9, 9)], 1028, [(0, 13)]),
    ([(10, 15), (17, 25), (11, 11), (7, 3), (9, 10), (15, 15), (16, 18), (2, 5), (10, 15)], 20, [(2, 5), (9, 19)]),
])
def test_container_from_ranges_container_rs_598_788(ranges, n, exp):
    assert _b(O.container_from_ranges(ranges, n)) == exp


def test_container_iter_isin_union_intersection_container_rs_790_925():
    cx = [11, 1, 2, 3, 20, 5, 0, 5, 4, 18, 6, 19, 1, 13, 20, 27]
    c1 = O.container_from_indices(cx, 20)
    assert O.container_expand(c1).tolist() == [0, 1, 2, 3, 4, 5, 6, 11, 13, 18, 19]      # :798-807
    assert O.container_isin(c1, 5) and not O.container_isin(c1, 12) and not O.container_isin(c1, 73)
    c2i = O.container_from_indices([11, 20, 5, 5, 4, 18, 6, 19, 13, 20, 27], 20)        # isin2 :866-874
    assert not O.container_isin(c2i, 1) and not O.container_isin(c2i, 3) and O.container_isin(c2i, 5)
    c2 = O.container_from_ranges([(10, 15), (17, 25), (11, 11), (7, 3), (9, 10), (15, 15), (16, 18), (2, 5), (10, 15)], 20)
    assert _b(O.container_union(c1, c2)) == [(0, 6), (9, 19)]                           # :877-901
    c3 = O.container_from_indices([13, 1, 2, 7, 5, 19, 21, 1, 9, 10, 11], 15)
    exp = [(1, 2), (5, 5), (11, 11), (13, 13)]
    assert _b(O.container_intersection(c1, c3)) == exp and _b(O.container_intersection(c3, c1)) == exp
    empty = np.zeros((0, 2), np.uint64)
    assert _b(O.container_intersection(c1, empty)) == [] and _b(O.container_intersection(empty, c1)) == []
    c4 = O.container_from_indices([7, 8, 9, 10, 12, 14], 15)
    assert _b(O.container_intersection(c1, c4)) == []


# ----------------------------------------------------------------------------- analysis.rs centres
def test_center_small_systems_analysis_rs_488_629():
    one = np.array([[4.5, 3.2, 1.7]], np.float32)
    c = O.estimate_center(one, [0], BOX10)
    assert_approx(c[0], 4.5); assert_approx(c[1], 3.2); assert_approx(c[2], 1.7)
    two = np.array([[4.5, 3.2, 1.7], [4.0, 2.8, 3.0]], np.float32)
    c = O.estimate_center(two, [0, 1], BOX10)
    assert_approx(c[0], 4.25); assert_approx(c[1], 3.0); assert_approx(c[2], 2.35)
    g, n = O.get_center(two, [0, 1], BOX10), O.center_naive(two, [0, 1])
    for a, b in zip(g, n):
        assert_approx(a, b)
    pbc = np.array([[4.5, 3.2, 1.7], [9.8, 9.5, 3.0]], np.float32)
    for fn in (O.estimate_center, O.get_center):
        c = fn(pbc, [0, 1], BOX10)
        assert_approx(c[0], 2.15); assert_approx(c[1], 1.35); assert_approx(c[2], 2.35)
    five = np.array([[3.3, 0.3, 2.5], [4.3, 1.2, 9.8], [3.2, 5.6, 0.5], [0.2, 9.0, 6.6], [8.7, 5.0, 2.4]], np.float32)
    oob = np.array([[3.3, 10.3, 2.5], [4.3, 1.2, -0.2], [13.2, 15.6, 0.5], [10.2, -1.0, 6.6], [-1.3, 5.0, 2.4]], np.float32)
    for pts in (five, oob):
        c = O.estimate_center(pts, range(5), BOX10)
        assert_approx(c[0], 2.634386, 1e-4); assert_approx(c[1], 9.775156, 1e-4); assert_approx(c[2], 1.1748, 1e-4)
        m = [10.3, 5.4, 3.8, 10.1, 7.6]                                                  # :932-989
        c = O.estimate_center(pts, range(5), BOX10, mass=m)
        assert_approx(c[0], 1.9526, 1e-4); assert_approx(c[1], 9.7567, 1e-4); assert_approx(c[2], 1.8812, 1e-4)


def test_com_two_atoms_analysis_rs_813_929():
    two = np.array([[4.5, 3.2, 1.7], [4.0, 2.8, 3.0]], np.float32)
    m = [12.8, 0.4]
    c = O.estimate_center(two, [0, 1], BOX10, mass=m)
    assert_approx(c[0], 4.485, 1e-4); assert_approx(c[1], 3.188, 1e-4); assert_approx(c[2], 1.73549, 1e-4)
    g, n = O.get_center(two, [0, 1], BOX10, mass=m), O.center_naive(two, [0, 1], mass=m)
    for a, b in zip(g, n):
        assert_approx(a, b, 1e-5)
    pbc = np.array([[4.5, 3.2, 1.7], [9.8, 9.5, 3.0]], np.float32)
    c = O.estimate_center(pbc, [0, 1], BOX10, mass=m)
    assert_approx(c[0], 4.4904, 1e-4); assert_approx(c[1], 3.1630, 1e-4); assert_approx(c[2], 1.7355, 1e-4)
    c = O.get_center(pbc, [0, 1], BOX10, mass=m)
    assert_approx(c[0], 4.35757, 1e-4); assert_approx(c[1], 3.08788, 1e-4); assert_approx(c[2], 1.7393947, 1e-4)


def test_center_real_system_analysis_rs_631_763(example):
    pos, box = example["pos"], example["box9"]
    mem = O.container_expand(example["blocks_Membrane"]); prot = O.container_expand(example["blocks_Protein"])
    cm, cp = O.center_naive(pos, mem), O.center_naive(pos, prot)
    assert_approx(cm[0], 6.47077, 1e-4); assert_approx(cm[1], 6.52237, 1e-4); assert_approx(cm[2], 5.77978, 1e-4)
    assert_approx(cp[0], 9.85718, 1e-4); assert_approx(cp[1], 2.46213, 1e-4); assert_approx(cp[2], 5.45931, 1e-4)
    gm, gp = O.get_center(pos, mem, box), O.get_center(pos, prot, box)
    assert_approx(gm[2], cm[2], 1e-4)
    for a, b in zip(gp, cp):
        assert_approx(a, b, 1e-4)
    # same mass for all atoms -> COM == centre (:991-1013)
    m = np.full(pos.shape[0], 12.3, np.float32)
    for idx, c in ((mem, gm), (prot, gp)):
        com = O.get_center(pos, idx, box, mass=m)
        for a, b in zip(com, c):
            assert_approx(a, b, 1e-4)


def test_center_errors_analysis_rs_651_741(example):
    pos, box = example["pos"].copy(), example["box9"]
    prot = O.container_expand(example["blocks_Protein"])
    pos[15, 0] = np.nan  # reset_position
    for fn in (lambda: O.get_center(pos, prot, box), lambda: O.estimate_center(pos, prot, box),
               lambda: O.center_naive(pos, prot)):
        with pytest.raises(O.OracleError) as e:
            fn()
        assert e.value.status == O.E_NO_POSITION and e.value.index == 15
    with pytest.raises(O.OracleError) as e:
        O.get_center(example["pos"], prot, None)
    assert e.value.status == O.E_NO_BOX


def test_naive_com_protein_tpr_masses_analysis_rs_1186_1200(example):
    pos = example["pos"]; m = np.full(pos.shape[0], np.nan, np.float32); m[:61] = example["protein_masses"]
    c = O.center_naive(pos, np.arange(61), mass=m)
    assert_approx(c[0], 9.85456, 1e-4); assert_approx(c[1], 2.44974, 1e-4); assert_approx(c[2], 5.51983, 1e-4)


def test_estimate_com_aa_analysis_rs_1016_1056(aa):
    pos, box, m = aa["pos"], aa["box9"], aa["masses"]
    pep = O.container_expand(aa["blocks_peptide"]); mem = O.container_expand(aa["blocks_membrane"])
    c = O.estimate_center(pos, pep, box, mass=m)
    assert_approx(c[0], 4.047723, 1e-4); assert_approx(c[1], 3.764632, 1e-4); assert_approx(c[2], 3.2633042, 1e-4)
    c = O.estimate_center(pos, mem, box, mass=m)
    assert_approx(c[0], 1.44719, 1e-4); assert_approx(c[1], 0.45375, 1e-4); assert_approx(c[2], 3.74161, 1e-4)
    g, n = O.get_center(pos, pep, box, mass=m), O.center_naive(pos, pep, mass=m)
    for a, b in zip(g, n):
        assert_approx(a, b)      # default margin (:1049-1051)
    gm, nm = O.get_center(pos, mem, box, mass=m), O.center_naive(pos, mem, mass=m)
    assert_approx(gm[2], nm[2])


# ----------------------------------------------------------------------------- analysis.rs distances
@pytest.mark.parametrize("dim,exp", [
    ("x", 6.3029766), ("y", -5.566175), ("z", -0.32046986), ("xy", 8.408913),
    ("xz", 6.311118), ("yz", 5.5753927), ("xyz", 8.415017), ("none", 0.0),
])
def test_group_distance_analysis_rs_1268_1354(example, dim, exp):
    pos, box = example["pos"], example["box9"]
    mem = O.container_expand(example["blocks_Membrane"]); prot = O.container_expand(example["blocks_Protein"])
    c1, c2 = O.get_center(pos, prot, box), O.get_center(pos, mem, box)   # analysis.rs:348-360
    assert_approx(O.distance(c1, c2, dim, box), exp, 1e-4)


def test_group_all_distances_analysis_rs_1420_1530(example):
    pos, box = example["pos"], example["box9"]
    mem = O.container_expand(example["blocks_Membrane"]); prot = O.container_expand(example["blocks_Protein"])
    n = prot.size
    d = O.group_all_distances(pos, prot, prot, "xyz", box)
    assert d.shape == (n, n) and np.all(np.diag(d) == 0.0)
    assert all(approx_f32(d[i, j], d[j, i]) for i in range(n) for j in range(n))
    assert_approx(d.max(), 4.597961); assert_approx(d[0, 1], 0.31040135)
    assert_approx(d[n - 1, 0], 4.266728); assert_approx(d[n - 1, n - 2], 0.31425142)
    d = O.group_all_distances(pos, prot, prot, "z", box)
    assert all(approx_f32(d[i, j], -d[j, i]) for i in range(n) for j in range(n))
    assert_approx(d.max(), 4.383, 1e-5); assert_approx(d.min(), -4.383, 1e-5)
    assert_approx(d[0, 1], 0.09, 1e-5); assert_approx(d[n - 1, 0], -4.213, 1e-5); assert_approx(d[n - 1, n - 2], -0.101, 1e-5)
    d = O.group_all_distances(pos, mem, prot, "xy", box)
    assert d.shape == (mem.size, n)
    assert_approx(d.max(), 9.190487, 1e-5); assert_approx(d.min(), 0.02607, 1e-5)
    assert_approx(d[0, 0], 3.747651); assert_approx(d[1240, 12], 3.7207017)
    assert_approx(d[12, 34], 6.2494035); assert_approx(d[6143, 60], 4.7850933)


def test_atoms_distance_analysis_rs_1595_1619(example):
    pos, box = example["pos"], example["box9"]
    n = pos.shape[0]
    assert_approx(O.distance(pos[0], pos[1], "xyz", box), 0.31040135)
    assert_approx(O.distance(pos[n - 1], pos[0], "xyz", box), 6.664787)
    assert_approx(O.distance(pos[n - 1], pos[n - 2], "xyz", box), 4.062491)


# ----------------------------------------------------------------------------- modifying.rs / utility.rs
def test_atoms_translate_modifying_rs_504_524(example):
    pos, box = example["pos"], example["box9"]
    out = O.translate(pos, np.arange(pos.shape[0]), [3.5, -1.1, 5.4], box)
    assert_approx(out[0, 0], 12.997); assert_approx(out[0, 1], 0.889); assert_approx(out[0, 2], 1.64453)
    assert_approx(out[-1, 0], 12.329); assert_approx(out[-1, 1], 10.086); assert_approx(out[-1, 2], 7.475)


def test_atoms_wrap_modifying_rs_688_735(example):
    pos, box = example["pos"], example["box9"]
    moved = pos.copy()
    moved[[154, 1754, 12345, 4, 37, 0]] += np.array([box[0] * np.float32(3.0), -box[1], 0.0], np.float32)
    moved[[13, 65, 9853, 16843, 7832, 489]] += np.array([0.0, box[1], -box[2] * np.float32(2.0)], np.float32)
    out = O.wrap_atoms(moved, np.arange(pos.shape[0]), box)
    assert np.all(np.abs(out - pos) <= 1e-5)


CENTER_EXPECT = {  # utility.rs:337-522 : dim -> (atom1 xyz, atom2 xyz)
    "none": ((9.497, 1.989, 7.498), (8.829, 11.186, 2.075)),
    "x": ((6.1465545, 1.989, 7.498), (5.478555, 11.186, 2.075)),
    "y": ((9.497, 6.033055, 7.498), (8.829, 2.2167444, 2.075)),
    "z": ((9.497, 1.989, 7.6634398), (8.829, 11.186, 2.2404397)),
    "xy": ((6.1465545, 6.033055, 7.498), (5.478555, 2.2167444, 2.075)),
    "xz": ((6.1465545, 1.989, 7.6634398), (5.478555, 11.186, 2.2404397)),
    "yz": ((9.497, 6.033055, 7.6634398), (8.829, 2.2167444, 2.2404397)),
    "xyz": ((6.1465545, 6.033055, 7.6634398), (5.478555, 2.2167444, 2.2404397)),
}


@pytest.mark.parametrize("dim", list(CENTER_EXPECT))
def test_atoms_center_utility_rs_337_522(example, dim):
    pos, box = example["pos"], example["box9"]
    prot = O.container_expand(example["blocks_Protein"])
    out = O.atoms_center(pos, prot, dim, box)
    a1, a2 = CENTER_EXPECT[dim]
    for k in range(3):
        assert_approx(out[0, k], a1[k]); assert_approx(out[-1, k], a2[k])
    c, bc = O.estimate_center(out, prot, box), O.box_center(box)
    for k, ax in enumerate("xyz"):
        if ax in dim:
            assert_approx(c[k], bc[k])


CENTER_MASS_EXPECT = {  # utility.rs:586-760
    "x": ((3.456437, 3.899, 4.993), (2.0444372, 3.823, 0.378)),
    "y": ((4.322, 3.475028, 4.993), (2.910, 3.399028, 0.378)),
    "z": ((4.322, 3.899, 5.4376106), (2.910, 3.823, 0.82261086)),
    "xy": ((3.456437, 3.475028, 4.993), (2.0444372, 3.399028, 0.378)),
    "xz": ((3.456437, 3.899, 5.4376106), (2.0444372, 3.823, 0.82261086)),
    "yz": ((4.322, 3.475028, 5.4376106), (2.910, 3.399028, 0.82261086)),
    "xyz": ((3.456437, 3.475028, 5.4376106), (2.0444372, 3.399028, 0.82261086)),
}


@pytest.mark.parametrize("dim", list(CENTER_MASS_EXPECT))
def test_atoms_center_mass_utility_rs_586_760(aa, dim):
    pos, box, m = aa["pos"], aa["box9"], aa["masses"]
    pep = O.container_expand(aa["blocks_peptide"])
    out = O.atoms_center(pos, pep, dim, box, mass=m)
    a1, a2 = CENTER_MASS_EXPECT[dim]
    for k in range(3):
        assert_approx(out[0, k], a1[k]); assert_approx(out[-1, k], a2[k])


# ----------------------------------------------------------------------------- rmsd.rs
P3 = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
THIRD = [0.3333333] * 3
ROT90 = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]).T   # nalgebra Matrix3::from lists COLUMNS


def test_kabsch_identity_rotation_translation_rmsd_rs_618_744():
    R, t, r = O.kabsch_rmsd(P3, P3, [1, 1, 1], THIRD, THIRD, 3.0)
    assert np.allclose(R, np.eye(3), atol=1e-6) and np.linalg.norm(t) <= 1e-6 and abs(r) <= 1e-6
    q = [[0.6666667, 1.0, 0.0], [-0.3333333, 0.0, 0.0], [0.6666667, 0.0, 1.0]]
    R, t, r = O.kabsch_rmsd(P3, q, [1, 1, 1], THIRD, THIRD, 3.0)
    assert np.allclose(R, ROT90, atol=1e-6) and np.linalg.norm(t) <= 1e-6 and abs(r) <= 1e-6
    q = [[2.0, 1.0, 1.0], [1.0, 2.0, 1.0], [1.0, 1.0, 2.0]]
    R, t, r = O.kabsch_rmsd(P3, q, [1, 1, 1], THIRD, [1.3333333] * 3, 3.0)
    assert np.allclose(R, np.eye(3), atol=1e-6) and np.allclose(t, [1, 1, 1], rtol=1e-6, atol=1e-6) and abs(r) <= 1e-6
    q = [[1.6666666, 2.0, 1.0], [0.6666666, 1.0, 1.0], [1.6666666, 1.0, 2.0]]
    R, t, r = O.kabsch_rmsd(P3, q, [1, 1, 1], THIRD, [1.3333333] * 3, 3.0)
    assert np.allclose(R, ROT90, atol=1e-6) and np.allclose(t, [1, 1, 1], rtol=1e-6, atol=1e-6) and abs(r) <= 1e-6


def test_kabsch_nonzero_rmsd_rs_746_780():
    p = [[4.3, 2.1, -5.2], [1.4, 2.1, 3.9], [2.4, -3.3, 1.8]]
    q = [[2.2, 0.0, 4.6], [-1.4, 0.2, 0.3], [1.3, 9.9, 11.3]]
    R, t, r = O.kabsch_rmsd(p, q, [1, 1, 1], [2.7, 0.3, 0.16666667], [0.7, 3.3666667, 5.4], 3.0)
    cols = np.array([[0.8842437, -0.10340805, -0.45543456], [0.2840647, -0.65496445, 0.70023507],
                     [-0.37070346, -0.7485511, -0.5497733]])
    exp = cols.T   # Matrix3::from([[c0],[c1],[c2]]) is column-major
    assert np.all(np.abs(R - exp) <= 1e-6 + 1e-6 * np.abs(exp)), (R, exp)
    assert np.allclose(t, [-2.0, 3.066666, 5.233333], rtol=1e-6, atol=1e-6)
    assert_approx(r, 4.471225, 1e-6)


RMSD_EXPECTED = [0.23669721, 0.2634763, 0.26021627, 0.21364464, 0.22166993, 0.19383307, 0.26422343,
                 0.27013618, 0.26398134, 0.23475659, 0.24208021]    # rmsd.rs:811-814
# The reference asserts these at 4 ulps against its own nalgebra f32 SVD.  The rotation maximises the
# UNWEIGHTED covariance while the RMSD sum is mass-weighted (rmsd.rs:567-570 vs :592-599), so the RMSD
# is not stationary in R and the f32 rounding of nalgebra's SVD shows up at first order (~1e-6
# relative).  The oracle computes R exactly (fp64), hence <= 3.3e-7 nm deviations; 5e-7 is the pin.
RMSD_EPS = 5e-7


def _short_system(example, short_traj):
    keep = short_traj["keep"].astype(np.int64)
    ref_pos = short_traj["gro_keep"]           # example.tpr positions == example.gro positions
    masses = np.full(keep.size, np.nan, np.float32)
    masses[:61] = example["protein_masses"]    # atoms outside the group never have their mass read
    return keep, ref_pos, masses


def test_calc_rmsd_trajectory_rmsd_rs_795_820(example, short_traj):
    keep, ref_pos, masses = _short_system(example, short_traj)
    sel = np.arange(61)
    ref = ref_pos.copy()
    ref[100, 0] = np.nan   # "should work even if we remove position of some atom that is not in the group"
    for f in range(11):
        r, _ = O.calc_rmsd(ref, masses, sel, example["box9"], short_traj["frames"][f], masses, sel, short_traj["boxes9"][f])
        assert_approx(r, RMSD_EXPECTED[f], RMSD_EPS, msg="frame %d" % f)
    r, _ = O.calc_rmsd(ref_pos, masses, sel, example["box9"], ref_pos, masses, sel, example["box9"])
    assert_approx(r, 0.0, 1e-4)                                                         # :783-792


def test_calc_rmsd_broken_at_pbc_rmsd_rs_843_866(example, short_traj):
    keep, ref_pos, masses = _short_system(example, short_traj)
    sel = np.arange(61); box = example["box9"]
    broken = O.translate(ref_pos, np.arange(ref_pos.shape[0]), [3.2, -2.1, -4.6], box)
    r1, _ = O.calc_rmsd(broken, masses, sel, box, ref_pos, masses, sel, box)
    r2, _ = O.calc_rmsd(ref_pos, masses, sel, box, broken, masses, sel, box)
    assert_approx(r1, 0.0, 1e-4); assert_approx(r2, 0.0, 1e-4)


def _cmp_wrapped(a, b, box, eps):
    d = np.abs(a - b)
    for k in range(3):
        dk = np.minimum(d[:, k], np.abs(d[:, k] - box[k]))
        assert dk.max() <= eps, (k, dk.max())


def test_rmsd_fit_same_and_rotated_rmsd_rs_900_948(example, short_traj):
    keep, ref_pos, masses = _short_system(example, short_traj)
    sel = np.arange(61); box = example["box9"]
    r, fitted = O.calc_rmsd_and_fit(ref_pos, masses, sel, box, ref_pos, masses, sel, box)
    assert_approx(r, 0.0, 1e-4); _cmp_wrapped(fitted, ref_pos, box, 1e-3)
    moved = O.translate(ref_pos, np.arange(ref_pos.shape[0]), [-1.1, 3.4, 2.7], box)
    rot = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]], np.float32)   # Matrix3::new is row-major
    moved = (rot @ moved.T).T.astype(np.float32)
    r, fitted = O.calc_rmsd_and_fit(ref_pos, masses, sel, box, moved, masses, sel, box)
    assert_approx(r, 0.0, 1e-4); _cmp_wrapped(fitted, ref_pos, box, 1e-3)


def test_rmsd_fit_trajectory_golden_files_rmsd_rs_950_1073(example, short_traj):
    """short_trajectory_fit.xtc / _broken_fit.xtc are byte-exact goldens in the reference; after decode
    they carry the xtc quantisation: the files are written at precision 100, i.e. every coordinate is
    rounded to the nearest 0.01 nm, so |fitted - decoded| <= 0.005 nm (+ f32 rounding)."""
    assert float(short_traj["precision"]) == 100.0
    # + 2e-4: the reference rotates with its f32-SVD R (error ~1e-5) and atoms sit up to ~10 nm from the centre
    tol = 0.5 / float(short_traj["precision"]) + 2e-4
    keep, ref_pos, masses = _short_system(example, short_traj)
    sel = np.arange(61); box = example["box9"]
    broken_ref = O.translate(ref_pos, np.arange(ref_pos.shape[0]), [3.2, -2.1, -4.6], box)
    for f in range(11):
        r, fitted = O.calc_rmsd_and_fit(ref_pos, masses, sel, box, short_traj["frames"][f], masses, sel, short_traj["boxes9"][f])
        assert_approx(r, RMSD_EXPECTED[f], RMSD_EPS)
        assert np.abs(fitted - short_traj["fit"][f]).max() <= tol, f
        r, fitted = O.calc_rmsd_and_fit(broken_ref, masses, sel, box, short_traj["frames"][f], masses, sel, short_traj["boxes9"][f])
        assert np.abs(fitted - short_traj["broken_fit"][f]).max() <= tol, f


def test_rmsd_errors_rmsd_rs_1075_1275(example, short_traj):
    keep, ref_pos, masses = _short_system(example, short_traj)
    sel = np.arange(61); box = example["box9"]
    with pytest.raises(O.OracleError) as e:
        O.calc_rmsd(ref_pos, masses, sel, box, ref_pos, masses, np.arange(60), box)
    assert e.value.status == O.E_INCONSISTENT_GROUP and e.value.counts == (61, 60)
    with pytest.raises(O.OracleError) as e:
        O.calc_rmsd(ref_pos, masses, sel, None, ref_pos, masses, sel, box)
    assert e.value.status == O.E_NO_BOX
    with pytest.raises(O.OracleError) as e:
        O.calc_rmsd(ref_pos, masses, np.zeros(0, np.uint64), box, ref_pos, masses, sel, box)
    assert e.value.status == O.E_EMPTY_GROUP
    bad = ref_pos.copy(); bad[7, 0] = np.nan
    with pytest.raises(O.OracleError) as e:
        O.calc_rmsd(ref_pos, masses, sel, box, bad, masses, sel, box)
    assert e.value.status == O.E_NO_POSITION and e.value.index == 7
    badm = masses.copy(); badm[9] = np.nan
    with pytest.raises(O.OracleError) as e:
        O.calc_rmsd(ref_pos, badm, sel, box, ref_pos, masses, sel, box)
    assert e.value.status == O.E_NO_MASS and e.value.index == 9
    # strict mode reproduces the reference's NotOrthogonal (system/mod.rs:1120-1124)
    tri = O.box_from_lengths_angles([13.0, 13.0, 11.0], [80.0, 70.0, 120.0])
    O.set_strict_orthogonal(True)
    try:
        with pytest.raises(O.OracleError) as e:
            O.calc_rmsd(ref_pos, masses, sel, tri, ref_pos, masses, sel, box)
        assert e.value.status == O.E_NOT_ORTHOGONAL
    finally:
        O.set_strict_orthogonal(False)
