"""GPU parity of geometry selection (gr_group_create_from_geometries, k_shape_mask) with the oracle and the reference's
known answers: System::group_create_from_geometry / _geometries (src/system/groups.rs:94-188, tests :1578-1668) over
Shape::inside (src/structures/shape.rs).  The result is an index list: it must be IDENTICAL."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "shape_cases.json")))


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def build(G, spec):
    k = spec["kind"]
    if k == "sphere": return G.Sphere(spec["position"], spec["radius"])
    if k == "rectangular": return G.Rectangular(spec["position"], *spec["size"])
    if k == "cylinder": return G.Cylinder(spec["position"], spec["radius"], spec["height"], G.Dimension[spec["orientation"].upper()])
    return G.TriangularPrism(spec["base1"], spec["base2"], spec["base3"], spec["height"])


def members(s, name):
    return np.array(list(s.group_container(name)), dtype=np.uint64)


@pytest.fixture()
def ex(G, example):
    s = G.System(example["pos"].shape[0], box=example["box9"], positions=example["pos"])
    for g in ("Protein", "Membrane", "W", "ION"):
        s.group_create_from_ranges(g, [tuple(b) for b in example["blocks_" + g]])
    yield s
    s.close()


@pytest.mark.parametrize("case", CASES["groups"], ids=lambda c: "%s_L%d" % (c["source"], c["line"]))
def test_reference_group_counts_on_example_gro(G, ex, example, case):
    shapes = [build(G, sp) for sp in case["shapes"]]
    ex.group_create_from_geometries("Selected", case["source"], shapes)
    assert ex.group_get_n_atoms("Selected") == case["count"]                       # the reference's pinned count
    want = O.group_from_geometries(example["pos"], O.container_expand(example["blocks_" + case["source"]]), example["box9"], case["shapes"])
    assert np.array_equal(members(ex, "Selected"), want)
    for i in members(ex, "Selected")[:50]:                                            # groups.rs:1591-1594
        assert all(sh.inside(example["pos"][int(i)], example["box9"]) for sh in shapes)


def test_several_shapes_naive_and_replacing_a_group(G, ex, example):
    sphere = {"kind": "sphere", "position": [0.5, 4.5, 3.5], "radius": 4.6}
    prism = {"kind": "prism", "base1": [8.0, 8.0, 8.0], "base2": [15.0, 12.0, 8.0], "base3": [9.5, 7.3, 8.0], "height": 5.4}
    rect = {"kind": "rectangular", "position": [2.0, 1.0, 1.5], "size": [6.0, 7.0, 5.0]}
    idx = O.container_expand(example["blocks_W"])
    assert ex.group_create_from_geometries("Sel", "W", [build(G, sphere), build(G, prism)]) is False
    assert np.array_equal(members(ex, "Sel"), O.group_from_geometries(example["pos"], idx, example["box9"], [sphere, prism]))
    # same name again: replaced, AlreadyExistsWarning (groups.rs:117, Groups::add)
    assert ex.group_create_from_geometries("Sel", "W", [build(G, sphere), build(G, rect)], naive=True) is True
    assert np.array_equal(members(ex, "Sel"), O.group_from_geometries(example["pos"], idx, example["box9"], [sphere, rect], naive=True))
    # a group built from a geometry is an ordinary group: usable as a source and in analyses
    ex.group_create_from_geometry("Sel", "W", build(G, sphere))
    ex.group_create_from_geometry("Sel2", "Sel", build(G, {"kind": "cylinder", "position": [5.0, 8.0, 3.0], "radius": 3.0, "height": 6.0, "orientation": "Y"}))
    sub = O.group_from_geometries(example["pos"], members(ex, "Sel"), example["box9"],
                                  [{"kind": "cylinder", "position": [5.0, 8.0, 3.0], "radius": 3.0, "height": 6.0, "orientation": "Y"}])
    assert np.array_equal(members(ex, "Sel2"), sub) and sub.size > 0
    np.testing.assert_allclose(ex.group_get_center("Sel2"), O.get_center(example["pos"], sub, example["box9"]), atol=1e-5, rtol=0)


def test_errors_follow_the_reference_order(G, ex, example):
    cyl = G.Cylinder([5.0, 8.0, 3.0], 2.0, 6.0, G.Dimension.Y)
    with pytest.raises(G.GroupError) as e:                                            # groups.rs:1674-1681
        ex.group_create_from_geometry("Selected Me>brane", "Membrane", cyl)
    assert e.value.variant == "InvalidName"
    with pytest.raises(G.GroupError) as e:                                            # :1683-1692
        ex.group_create_from_geometry("Selected Membrane", "brane", cyl)
    assert e.value.variant == "InvalidQuery"
    with pytest.raises(G.GroanError):                                                 # no NaiveShape for the prism
        ex.group_create_from_geometry("P", "W", G.TriangularPrism([8, 8, 8], [15, 12, 8], [9.5, 7.3, 8], 5.4), naive=True)
    tric = np.array([13.0, 13.0, 11.0, 0, 0, 1.0, 0, 0, 0], np.float32)
    ex.set_box(tric)
    ex.set_strict_orthogonal(True)            # the reference's gate; without it a non-orthogonal box takes the extension
    with pytest.raises(G.GroupError) as e:                                            # groups.rs:108-110
        ex.group_create_from_geometry("S", "Membrane", cyl)
    assert e.value.variant == "InvalidSimBox" and e.value.detail.variant == "NotOrthogonal"
    ex.set_strict_orthogonal(False)
    ex.reset_box()
    with pytest.raises(G.GroupError) as e:                                            # :1694-1704
        ex.group_create_from_geometry("S", "Membrane", cyl)
    assert e.value.variant == "InvalidSimBox" and e.value.detail.variant == "DoesNotExist"
    assert not ex.group_exists("S")


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_on_synthetic_systems(G, seed):
    """uniform atoms in an orthorhombic cell (some far outside it, a few without position), random shapes of every kind,
    contiguous and scattered source groups: index lists identical to the oracle's"""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(50_000, 300_000))
    box = np.zeros(9, np.float32); box[:3] = rng.uniform(5.0, 14.0, 3)
    pos = (rng.uniform(-0.7, 1.7, (n, 3)) * box[:3]).astype(np.float32)
    pos[rng.integers(0, n, 40), 0] = np.nan                                           # Option<Vector3D>::None
    s = G.System(n, box=box, positions=pos)
    s.group_create_from_ranges("blockA", [(1000, n - 777)])
    scattered = np.unique(rng.integers(0, n, n // 3))
    s.group_create_from_indices("scattered", scattered)
    L = box[:3]
    def rnd_shape():
        k = rng.integers(0, 4)
        c = (rng.uniform(-0.2, 1.2, 3) * L).tolist()
        if k == 0: return {"kind": "sphere", "position": c, "radius": float(rng.uniform(0.3, 0.6) * L.min())}
        if k == 1: return {"kind": "rectangular", "position": c, "size": (rng.uniform(0.2, 1.1, 3) * L).tolist()}
        if k == 2: return {"kind": "cylinder", "position": c, "radius": float(rng.uniform(0.2, 0.6) * L.min()), "height": float(rng.uniform(0.2, 1.1) * L.min()),
                           "orientation": "xyz"[rng.integers(0, 3)]}
        ax = int(rng.integers(0, 3)); b = [(rng.uniform(0.0, 1.0, 3) * L) for _ in range(3)]
        for v in b: v[ax] = b[0][ax]
        return {"kind": "prism", "base1": b[0].tolist(), "base2": b[1].tolist(), "base3": b[2].tolist(), "height": float(rng.uniform(0.2, 0.9) * L[ax])}
    for trial in range(8):
        specs = [rnd_shape() for _ in range(int(rng.integers(1, 4)))]
        naive = bool(trial % 4 == 3) and all(sp["kind"] != "prism" for sp in specs)
        for src, idx in (("all", np.arange(n)), ("blockA", np.arange(1000, n - 776)), ("scattered", scattered)):
            s.group_create_from_geometries("picked", src, [build(G, sp) for sp in specs], naive=naive)
            want = O.group_from_geometries(pos, idx, box, specs, naive=naive)
            got = members(s, "picked")
            assert np.array_equal(got, want), (trial, src, specs, got.size, want.size)
    s.close()


TRIC_CELLS = {"triclinic_75_80_70": ([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]), "dodecahedron": ([6.0, 6.0, 6.0], [60.0, 60.0, 90.0]),
              "octahedron": ([6.0, 6.0, 6.0], [70.53, 109.47, 70.53]), "skewed_negative": ([6.5, 7.5, 6.0], [100.0, 95.0, 110.0]),
              # a flat cell, given as the box itself: the images a body can reach include |k| = 3 (rounds 1-3 tested |i|, |j|, |k| <= 2 only)
              "flat": np.array([12.0, 11.0, 3.5, 0, 0, 3.0, 0, -4.0, 4.5], np.float32)}


@pytest.mark.parametrize("cell", sorted(TRIC_CELLS))
def test_shapes_in_non_orthogonal_boxes(G, cell):
    """NEXT-2 as SURVEY section 8(f) words it (triclinic too): group_create_from_geometries / filter_geometry in triclinic,
    dodecahedral and octahedral cells.  The reference needs an orthogonal box here, so this is the library's extension
    (some lattice image of the atom lies inside the shape): index lists identical to the oracle's restatement of that
    definition, and -- independently of both -- equal to an fp64 search over 7 x 7 x 7 lattice images for every atom that is
    not within 2e-5 nm of a shape's surface."""
    from groan_rs_amd import workload as W
    box = TRIC_CELLS[cell] if cell == "flat" else W.box_from_lengths_angles(*TRIC_CELLS[cell])
    L = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    rng = np.random.default_rng(len(cell) * 7 + 5)
    n = 30000
    pos = (rng.uniform(-0.3, 1.3, (n, 3)) @ L).astype(np.float32)            # in and around the cell
    s = G.System(n, box=box, positions=pos)
    s.group_create_from_indices("Odd", list(range(1, n, 2)))
    specs = [dict(kind="sphere", position=[1.0, 5.5, 0.4], radius=1.7),
             dict(kind="rectangular", position=[5.5, 0.5, 4.8], size=[2.5, 1.5, 2.0]),
             dict(kind="cylinder", position=[0.3, 0.2, 5.0], radius=1.4, height=2.5, orientation="Z"),
             dict(kind="cylinder", position=[6.0, 3.0, 1.0], radius=1.1, height=3.0, orientation="X"),
             dict(kind="prism", base1=[1.0, 1.0, 5.0], base2=[3.5, 1.5, 5.0], base3=[2.0, 3.5, 5.0], height=2.2)]
    shapes = [G.Sphere(specs[0]["position"], 1.7), G.Rectangular(specs[1]["position"], 2.5, 1.5, 2.0),
              G.Cylinder(specs[2]["position"], 1.4, 2.5, G.Dimension.Z), G.Cylinder(specs[3]["position"], 1.1, 3.0, G.Dimension.X),
              G.TriangularPrism(specs[4]["base1"], specs[4]["base2"], specs[4]["base3"], 2.2)]
    R = 5 if cell == "flat" else 3
    images = np.array([(i, j, k) for i in range(-R, R + 1) for j in range(-R, R + 1) for k in range(-R, R + 1)], np.float64) @ L
    for spec, shape in zip(specs, shapes):
        for src, idx in (("all", np.arange(n)), ("Odd", np.arange(1, n, 2))):
            s.group_create_from_geometry("Sel", src, shape)
            got = members(s, "Sel")
            want = O.group_from_geometries(pos, idx, box, [spec])
            assert np.array_equal(got, want) and 0 < got.size < idx.size, (cell, spec["kind"], got.size, want.size)
            it = s.group_iter(src).filter_geometry(shape)                       # the iterator-level entry point: the same atoms
            assert list(it) == [int(i) for i in want]
        if spec["kind"] == "prism":
            continue
        anchor = np.asarray(spec["position"], np.float64)
        inside = np.zeros(n, bool); near = np.zeros(n, bool)
        step = 500 if cell == "flat" else 2000
        for a0 in range(0, n, step):
            e = pos[a0:a0 + step, None, :].astype(np.float64) - anchor + images[None, :, :]
            # vectorised per shape kind
            if spec["kind"] == "sphere":
                r = np.linalg.norm(e, axis=2); ins = r < spec["radius"]; dist = np.abs(r - spec["radius"])
            elif spec["kind"] == "rectangular":
                m = np.minimum(e, np.asarray(spec["size"]) - e).min(axis=2); ins = m >= 0; dist = np.abs(m)
            else:
                ax = "xyz".index(spec["orientation"].lower())
                along = e[:, :, ax]; planar = np.linalg.norm(np.delete(e, ax, axis=2), axis=2)
                m = np.minimum(np.minimum(along, spec["height"] - along), spec["radius"] - planar); ins = m >= 0; dist = np.abs(m)
            inside[a0:a0 + step] = ins.any(axis=1); near[a0:a0 + step] = (dist.min(axis=1) < 2e-5)
        s.group_create_from_geometry("Sel", "all", shape)
        got = np.zeros(n, bool); got[members(s, "Sel")] = True
        assert np.array_equal(got[~near], inside[~near]), (cell, spec["kind"], int((got[~near] != inside[~near]).sum()))
    s.close()
