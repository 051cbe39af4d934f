"""The iterator-level surface (AtomIterator mirror over the C ABI's anonymous-selection entry points gr_sel_*) against the
reference's own iterator tests (src/structures/iterators.rs:1699-2135) and the oracle.

Known answers replayed: iterator_estimate_center (:1908-1926), iterator_get_center (:1928-1946), the *_empty cases (NaN, not an
error: :1948-1962,2042-2054,2071-2083), iterator_estimate_com / iterator_get_com on the all-atom membrane (:1992-2040),
filter_geometry_immutable (:1699-1748: the same atoms as group_create_from_geometry), iterator_translate (:2086-2106),
iterator_wrap (:2108-2135).  Fixtures: tests/golden/example.npz, example_names.npz, aa_peptide.npz."""
import os
import types

import numpy as np
import pytest

import oracle_lib as O
from conftest import assert_approx

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def blocks(a):
    return [(int(x), int(y)) for x, y in a]


@pytest.fixture()
def example_system(G, example):
    s = G.System(example["pos"].shape[0], box=example["box9"], positions=example["pos"])
    for k in ("Protein", "Membrane", "W", "ION"):
        s.group_create_from_ranges(k, blocks(example["blocks_" + k]))
    yield s
    s.close()


def test_iterator_estimate_and_get_center_known_answers(G, example_system):
    s = example_system
    sphere = G.Sphere(s.group_estimate_center("Protein"), 2.0)
    c = s.group_iter("Membrane").filter_geometry(sphere).estimate_center()
    for got, want in zip(c, (9.8453, 2.4803874, 5.434977)):                     # iterators.rs:1923-1925
        assert_approx(got, want, epsilon=2e-6)
    sphere = G.Sphere(s.group_get_center("Protein"), 2.0)
    it = s.group_iter("Membrane").filter_geometry(sphere)
    c = it.get_center()
    for got, want in zip(c, (9.848716, 2.4805717, 5.4309845)):                  # :1943-1945
        assert_approx(got, want, epsilon=2e-6)
    # the same selection as an explicit container union / intersection (OrderedAtomIterator, :1563-1604)
    both = s.group_iter("Membrane").intersection(it).union(s.group_iter("Protein").intersection(it))
    assert both.container == it.container and len(both) == len(it)


def test_empty_iterators_give_nan_not_errors(G, example_system):
    s = example_system
    empty = s.container_iter(G.AtomContainer([]))
    for fn in (empty.get_center, empty.get_com, empty.estimate_center, empty.estimate_com, empty.get_center_naive, empty.get_com_naive):
        assert np.isnan(fn()).all()                                             # :1948-1962, 2042-2054, 2071-2083
    empty.translate([1.0, 2.0, 3.0]); empty.wrap()                              # nothing to do, no error
    # ... but the box is still checked first (simbox_check, :1153)
    s.reset_box()
    with pytest.raises(G.AtomError) as e:
        empty.get_center()
    assert e.value.variant == "InvalidSimBox"
    assert np.isnan(empty.get_center_naive()).all()                             # the naive centres need no box


def test_filter_geometry_matches_group_create_from_geometry(G, example_system, example):
    s = example_system
    pos, box, n = example["pos"], example["box9"], example["pos"].shape[0]
    shapes = {"Sphere": G.Sphere([10.5, 11.2, 1.7], 4.0), "Cylinder": G.Cylinder([0.5, 1.2, 10.3], 2.5, 4.5, G.Dimension.Z),
              "Rectangular": G.Rectangular([1.3, 12.4, 10.7], 6.5, 4.5, 5.0)}
    for name, shape in shapes.items():                                          # :1699-1748
        s.group_create_from_geometry(name, "all", shape)
        it = s.atoms_iter().filter_geometry(shape)
        assert it.container == s.group_container(name) and len(it) == s.group_get_n_atoms(name) > 0
    # NaiveShape through the iterator (:1802-1853) == the oracle's naive predicate
    it = s.group_iter("Membrane").filter_geometry_naive(shapes["Sphere"])
    want = O.group_from_geometries(pos, np.arange(61, 6205), box, [dict(kind="sphere", position=[10.5, 11.2, 1.7], radius=4.0)], naive=True)
    assert list(it) == [int(i) for i in want]


def test_iterator_com_on_the_all_atom_membrane(G, aa):
    n = aa["pos"].shape[0]
    s = G.System(n, masses=aa["masses"], box=aa["box9"], positions=aa["pos"])
    s.group_create_from_ranges("Peptide", blocks(aa["blocks_peptide"])); s.group_create_from_ranges("Membrane", blocks(aa["blocks_membrane"]))
    sphere = G.Sphere(s.group_get_center("Peptide"), 1.0)
    it = s.group_iter("Membrane").filter_geometry(sphere)
    for got, want in zip(it.estimate_com(), (3.985978, 3.7461767, 3.3526845)):  # iterators.rs:2011-2013
        assert_approx(got, want, epsilon=2e-6)
    # (the reference sums the filtered atoms sequentially in f32; the kernels sum in double: 5e-6 nm apart here, inside the 1e-5 nm bar)
    for got, want in zip(it.get_com(), (3.9912941, 3.744326, 3.3532307)):       # :2035-2037
        assert_approx(got, want, epsilon=1e-5)
    idx = np.array(list(it))
    np.testing.assert_allclose(it.get_com_naive(), O.center_naive(aa["pos"], idx, mass=aa["masses"]), atol=1e-5, rtol=0)
    # pair distances between two anonymous selections == the group call on the same atoms
    a, b = s.container_iter(G.AtomContainer([(0, 40)])), it
    d = a.all_distances(b, G.Dimension.XYZ)
    np.testing.assert_allclose(d, O.group_all_distances(aa["pos"], np.arange(41), idx, "xyz", aa["box9"]), atol=1e-5, rtol=0)
    s.close()


def test_iterator_translate_and_wrap(G, example):
    d = np.load(os.path.join(HERE, "golden", "example_names.npz"))
    st = types.SimpleNamespace(n_atoms=int(d["resid"].size), resid=d["resid"], atomid=d["atomid"],
                               resname=[x.decode() for x in d["resname"]], atomname=[x.decode() for x in d["atomname"]])
    pos, box, n = example["pos"], example["box9"], example["pos"].shape[0]
    s = G.System(n, box=box, positions=pos)
    ala = s.selection_iter("resname ALA", st)
    ala.translate([3.5, -1.1, 5.4])                                             # iterators.rs:2086-2106
    got = s.get_positions()
    for a, want in ((31, (0.23069, 1.567, 10.745)), (52, (0.28168964, 1.231, 9.237))):
        for g_, w in zip(got[a], want):
            assert_approx(g_, w, epsilon=2e-6)
    idx = np.array(list(ala))
    np.testing.assert_array_equal(got, O.translate(pos, idx, [3.5, -1.1, 5.4], box))        # bit for bit, everything else untouched
    # iterator_wrap (:2108-2135): move everything 1000 nm away without PBC, wrap the alanines only
    far = (pos + np.float32(1000.0)).astype(np.float32)
    s.set_frame(far, box)
    ala.wrap()
    got = s.get_positions()
    inside = np.zeros(n, bool); inside[idx] = True
    assert (got[inside] <= box[:3]).all() and (got[inside] >= 0).all() and (got[~inside] >= 1000.0).all()
    # 77 box lengths away: the reference's loop rounds once per turn (and drifts by an ulp of 1000 nm per turn); beyond 16 turns
    # the kernels take the closed form (DESIGN.md "wrap / min-image") -- same image, sub-ulp-of-the-input differences
    want = O.wrap_atoms(far, idx, box)
    np.testing.assert_array_equal(got[~inside], want[~inside])
    np.testing.assert_allclose(got[inside], want[inside], atol=2e-3, rtol=0)
    # an atom without position is the reference's AtomError::InvalidPosition(index), reported by the iterator itself
    bad = far.copy(); bad[int(idx[3])] = np.nan
    s.set_frame(bad, box)
    with pytest.raises(G.AtomError) as e:
        ala.wrap()
    assert e.value.variant == "InvalidPosition" and e.value.detail == int(idx[3])
    # a container that reaches outside the system never gets to a kernel
    with pytest.raises(G.AtomError) as e:
        s.container_iter(G.AtomContainer([(n - 2, n + 5)])).get_center_naive()
    assert e.value.variant == "OutOfRange"
    s.close()
