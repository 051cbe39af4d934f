"""Fused reducers of the pair-distance kernels (gr_group_all_distances_reduce: SURVEY 8 A8's extension; the reference's consumers take
the maximum and minimum of System::group_all_distances, src/system/analysis.rs:401-427 and :1420-1451): min / max / count-below / histogram
of the matrix, per row or over all of it, WITHOUT the matrix in memory.  Parity: every result EQUALS the same reduction of the full matrix
the plain call returns (bit for bit: the same tiles compute the same distances; min / max / counts do not depend on the order) and of the
oracle's matrix; the reference's pinned maxima / minima on example.gro; three kinds of cell, every dimension, gathered selections, a
batch of frames, BASELINE configs[2]'s 1e4 x 1e4 shape, and the reference's error order."""
import numpy as np
import pytest

import oracle_lib as O
from groan_rs_amd import workload as W

pytestmark = pytest.mark.gpu
DIMS = ["X", "Y", "Z", "XY", "XZ", "YZ", "XYZ"]


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def reductions_of(m, cut, nbins, rmax):
    """the reference-side consumer: numpy over the full matrix (f32 values compared as f32)"""
    scale = np.float32(nbins) / np.float32(rmax)
    fb = (m.astype(np.float32) * scale).astype(np.float32)
    ok = (m >= 0) & (fb < np.float32(nbins))
    return {"min": m.min(), "max": m.max(), "min_row": m.min(axis=1), "max_row": m.max(axis=1), "count": np.uint64((m < np.float32(cut)).sum()),
            "count_row": (m < np.float32(cut)).sum(axis=1).astype(np.uint64), "hist": np.bincount(fb[ok].astype(np.int64), minlength=nbins).astype(np.uint64)}


def check_all(G, s, g1, g2, dim, cut, nbins, rmax, slot=0, oracle=None):
    m = s.group_all_distances(g1, g2, G.Dimension[dim], slot=slot)
    if oracle is not None:
        np.testing.assert_allclose(m, oracle, atol=2e-6, rtol=0)
    want = reductions_of(m, cut, nbins, rmax)
    kw = dict(dim=G.Dimension[dim], first_slot=slot, n_frames=1)
    got, st = s.group_all_distances_reduce(g1, g2, "min", **kw); assert st[0] == 0 and got.shape == (1, 1) and got[0, 0] == want["min"], (dim, got, want["min"])
    got, _ = s.group_all_distances_reduce(g1, g2, "max", **kw); assert got[0, 0] == want["max"], (dim, got, want["max"])
    got, _ = s.group_all_distances_reduce(g1, g2, "min", per_row=True, **kw); assert np.array_equal(got[0], want["min_row"]), dim
    got, _ = s.group_all_distances_reduce(g1, g2, "max", per_row=True, **kw); assert np.array_equal(got[0], want["max_row"]), dim
    got, _ = s.group_all_distances_reduce(g1, g2, "count_below", param=cut, **kw); assert got.dtype == np.uint64 and got[0, 0] == want["count"], (dim, got, want["count"])
    got, _ = s.group_all_distances_reduce(g1, g2, "count_below", param=cut, per_row=True, **kw); assert np.array_equal(got[0], want["count_row"]), dim
    got, _ = s.group_all_distances_reduce(g1, g2, "hist", param=rmax, nbins=nbins, **kw); assert np.array_equal(got[0], want["hist"]), (dim, got[0][:8], want["hist"][:8])
    assert want["hist"].sum() > 0
    return m


@pytest.mark.parametrize("cell", ["ortho", "tric", "dodeca"])
def test_every_reduction_equals_the_reduction_of_the_matrix(G, cell):
    rng = np.random.default_rng(5)
    box = {"ortho": W.box_from_lengths_angles([6.5, 7.25, 5.0], [90.0, 90.0, 90.0]), "tric": W.box_from_lengths_angles([7.5, 7.0, 6.5], [75.0, 80.0, 70.0]), "dodeca": W.c4_box(7.0)}[cell]
    n = 6001
    pos = O.wrap_atoms((rng.random((n, 3)) * 9.0 - 1.0).astype(np.float32), np.arange(n), box)
    s = G.System(n, n_slots=1)
    s.set_frame(pos, box, slot=0)
    s.group_create_from_ranges("A", [(3, 1301)])                                  # 1299 rows: ragged against the 8-row tiles
    s.group_create_from_ranges("B", [(700, 5999)])                                # 5300 columns: five and a bit workgroup tiles, overlapping A
    s.group_create_from_indices("C", np.unique(rng.integers(0, n, 777)))          # a gathered selection
    ia, ib = np.arange(3, 1302), np.arange(700, 6000)
    for dim in DIMS:
        oracle = O.group_all_distances(pos, ia, ib, dim.lower(), box) if dim in ("XYZ", "X") else None
        check_all(G, s, "A", "B", dim, 1.5 if len(dim) > 1 else 0.1, 64, 4.0, oracle=oracle)
    check_all(G, s, "C", "A", "XYZ", 0.8, 4096, 5.0)
    check_all(G, s, "A", "A", "XYZ", 0.5, 100, 3.0)                               # a group with itself: the diagonal's zeros are entries of the matrix
    s.close()


def test_reference_maxima_and_minima_on_example_gro(G, example):
    """analysis.rs:1420-1451: the reference's own tests take max 4.597961 (Protein x Protein, XYZ) and, for Membrane x Protein in XY, max
    9.190487 / min 0.02607 of the matrix: here they come out of the fused reducers"""
    s = G.System(example["pos"].shape[0], box=example["box9"], positions=example["pos"])
    for name in ("Protein", "Membrane"):
        s.group_create_from_ranges(name, [tuple(int(x) for x in b) for b in example["blocks_" + name]])
    got, _ = s.group_all_distances_reduce("Protein", "Protein", "max")
    assert abs(float(got[0, 0]) - 4.597961) <= 1e-5
    mx, _ = s.group_all_distances_reduce("Membrane", "Protein", "max", dim=G.Dimension.XY)
    mn, _ = s.group_all_distances_reduce("Membrane", "Protein", "min", dim=G.Dimension.XY)
    assert abs(float(mx[0, 0]) - 9.190487) <= 1e-5 and abs(float(mn[0, 0]) - 0.02607) <= 1e-5
    s.close()


def test_batch_of_frames_statuses_and_errors(G):
    rng = np.random.default_rng(9)
    n, nf = 3000, 5
    box = W.box_from_lengths_angles([5.0, 5.5, 6.0], [90.0, 90.0, 90.0])
    s = G.System(n, n_slots=nf)
    frames = [(rng.random((n, 3)) * box[:3]).astype(np.float32) for _ in range(nf)]
    frames[3][1500] = np.nan                                                      # an atom of B without position: that frame fails, the others do not
    for f in range(nf):
        s.set_frame(frames[f], box, slot=f)
    s.group_create_from_ranges("A", [(0, 999)]); s.group_create_from_ranges("B", [(1000, 2999)])
    got, st = s.group_all_distances_reduce("A", "B", "min", per_row=True, first_slot=0, n_frames=nf, raise_on_error=False)
    assert list(st) == [0, 0, 0, 6, 0]                                            # GR_E_NO_POSITION
    cnt, st2 = s.group_all_distances_reduce("A", "B", "count_below", param=1.0, first_slot=0, n_frames=nf, raise_on_error=False)
    for f in (0, 1, 2, 4):
        m = s.group_all_distances("A", "B", slot=f)
        assert np.array_equal(got[f], m.min(axis=1)) and cnt[f, 0] == (m < np.float32(1.0)).sum()
    with pytest.raises(G.GroanError):
        s.group_all_distances_reduce("A", "nope", "min")
    with pytest.raises(G.GroanError):
        s.group_all_distances_reduce("A", "B", "hist", param=2.0, nbins=5000)     # more bins than the kernel's table
    with pytest.raises(G.GroanError):
        s.group_all_distances_reduce("A", "B", "hist", param=2.0, nbins=16, per_row=True)
    s.close()


def test_config3_shape_without_the_400_MB(G):
    """BASELINE configs[2]: 1e6 atoms in the triclinic cell, the first 1e4 against themselves, XYZ -- minimum over the off-diagonal is not asked
    for by the reference; its consumers take max and min of the whole matrix: equal to the reduction of the 400 MB matrix"""
    n, S = 1_000_000, 10_000
    box = W.box_from_lengths_angles([24.0, 23.0, 22.0], [75.0, 80.0, 70.0])
    s = G.System(n, n_slots=1)
    s.synth_uniform(0, box, 20260424)
    s.group_create_from_ranges("S", [(0, S - 1)])
    m = s.group_all_distances("S", "S", slot=0)
    mx, _ = s.group_all_distances_reduce("S", "S", "max")
    mxr, _ = s.group_all_distances_reduce("S", "S", "max", per_row=True)
    cnt, _ = s.group_all_distances_reduce("S", "S", "count_below", param=1.2)
    hist, _ = s.group_all_distances_reduce("S", "S", "hist", param=12.0, nbins=240)
    want = reductions_of(m, 1.2, 240, 12.0)
    assert mx[0, 0] == want["max"] and np.array_equal(mxr[0], want["max_row"]) and cnt[0, 0] == want["count"] and np.array_equal(hist[0], want["hist"])
    s.close()
