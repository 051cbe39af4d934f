"""The all-pairs matrix of a group WITH ITSELF (BASELINE configs[2]: "min-image pair distances for a 10k-atom selection";
reference: System::group_all_distances(group, group, dim), src/system/analysis.rs:401-427) takes the symmetric kernel
(k_pairdist_sym: tiles on and above the diagonal are computed, every tile above it is also written transposed).  It must give the
bits of the element-by-element kernel -- GR_TUNE_PAIRDIST_SYMMETRIC = 0 -- in every Dimension, which in turn is checked against
the oracle: triclinic and dodecahedral cells (orthorhombic cells stay on the plain kernel: they are bound by the stores), atoms far
outside the cell, Dimension::None (generic path: the mirror is computed, not assumed), sizes that are no multiple of the 64-atom
tile or of 4, gathered selections, atoms without a position, batches."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
DIMS = ["X", "Y", "Z", "XY", "XZ", "YZ", "XYZ", "NONE"]


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def both(G, s, group, dim, slot=0):
    s.set_tuning(pairdist_symmetric=1)
    a = s.group_all_distances(group, group, G.Dimension[dim], slot=slot)
    s.set_tuning(pairdist_symmetric=0)
    b = s.group_all_distances(group, group, G.Dimension[dim], slot=slot)
    s.set_tuning(pairdist_symmetric=1)
    return a, b


BOXES = {"ortho": ([6.5, 7.25, 5.0], [90.0, 90.0, 90.0]), "tric": ([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]), "dodeca": ([6.0, 6.0, 6.0], [60.0, 60.0, 90.0])}


@pytest.mark.parametrize("cell", list(BOXES))
@pytest.mark.parametrize("n", [256, 257, 1000, 1283, 2048])   # (below 256 atoms the plain kernel runs either way)
def test_self_matrix_is_the_bits_of_the_plain_kernel_and_the_oracle(G, cell, n):
    rng = np.random.default_rng(n)
    box = O.box_from_lengths_angles(*BOXES[cell])
    pos = (rng.random((n + 7, 3)) @ np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)).astype(np.float32) if cell != "ortho" \
        else (rng.random((n + 7, 3)) * box[:3]).astype(np.float32)
    pos[3] = pos[200]                                           # coincident atoms
    s = G.System(n + 7, n_slots=1)
    s.set_frame(pos, box, slot=0)
    s.group_create_from_ranges("S", [(5, n + 4)])               # (not aligned to the slot's 4-atom groups)
    idx = np.arange(5, n + 5)
    for dim in DIMS:
        a, b = both(G, s, "S", dim)
        assert a.shape == (n, n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (cell, n, dim, np.argwhere(a != b)[:4])
        if dim == "NONE":
            continue
        want = O.group_all_distances(pos, idx, idx, dim.lower(), box)
        if cell == "ortho" and len(dim) == 1:
            assert np.array_equal(a, want), dim
        elif cell == "ortho" or dim == "XYZ":
            np.testing.assert_allclose(a, want, atol=2e-6, rtol=0, err_msg=dim)
        else:
            # components of the 3-D minimum image: a pair whose two best images tie to rounding may pick either one.  Every entry that
            # differs from the oracle's is therefore checked on its own against an fp64 search over 7 x 7 x 7 lattice images that
            # shares nothing with the library or the oracle: the entry must be the requested component(s) of SOME image whose length
            # is within 1e-5 nm of the shortest one (round 3 tolerated 0.1 % of the entries without looking at them)
            off = np.argwhere(np.abs(a - want) > 2e-6)
            assert off.shape[0] <= a.size // 200, (dim, off.shape[0])
            if off.shape[0]:
                Lm = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
                rng3 = np.arange(-3, 4)
                imgs = np.array([(i, j, k) for i in rng3 for j in rng3 for k in rng3], np.float64) @ Lm
                p64 = pos.astype(np.float64)
                axes = ["XYZ".index(ch) for ch in dim]
                for (i, j) in off:
                    e = p64[idx[i]] - p64[idx[j]] + imgs                       # distance(x_i, x_j): the vector from j to i, every image
                    ln = np.linalg.norm(e, axis=1)
                    near = e[ln <= ln.min() + 1e-5]
                    val = near[:, axes[0]] if len(axes) == 1 else np.linalg.norm(near[:, axes], axis=1)
                    assert near.shape[0] >= 2 and np.abs(val - float(a[i, j])).min() <= 2e-6, (dim, int(i), int(j), float(a[i, j]), float(want[i, j]), val, ln.min())
        if len(dim) == 1:
            assert np.array_equal(a, -a.T), dim                 # signed: D[j][i] = -D[i][j]
        else:
            assert np.array_equal(a, a.T), dim
    s.close()


def test_self_matrix_with_far_atoms_gathered_selection_and_missing_positions(G):
    rng = np.random.default_rng(8)
    n = 3000
    box = O.box_from_lengths_angles([4.0, 5.0, 6.0], [70.0, 80.0, 75.0])
    pos = (rng.random((n, 3)) * box[:3]).astype(np.float32)
    pos[5] += np.float32([8.0, -10.0, 12.0]); pos[2000] -= np.float32([12.0, 0, 12.0]); pos[2999, 1] = 27.3     # far outside the cell
    s = G.System(n, n_slots=1)
    s.set_frame(pos, box, slot=0)
    members = np.concatenate([np.arange(0, 1500, 2), np.arange(1501, 3000, 3)])
    s.group_create_from_indices("Gathered", members.tolist())
    s.group_create_from_ranges("All", [(0, n - 1)])
    for group, idx in (("Gathered", members), ("All", np.arange(n))):
        for dim in ("X", "YZ", "XYZ"):
            a, b = both(G, s, group, dim)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (group, dim)
            if dim == "XYZ":
                # (atoms up to 27 nm out: one ulp of such a coordinate is 1.9e-6 nm)
                np.testing.assert_allclose(a, O.group_all_distances(pos, idx, idx, "xyz", box), atol=5e-6, rtol=0, err_msg=group)
    # an atom without a position: the same error, the same index, from both kernels
    bad = pos.copy(); bad[1234] = np.nan
    s.set_frame(bad, box, slot=0)
    for sym in (1, 0):
        s.set_tuning(pairdist_symmetric=sym)
        with pytest.raises(G.GroupError) as e:
            s.group_all_distances("All", "All", G.Dimension.XYZ)
        assert e.value.variant == "InvalidPosition" and e.value.detail == 1234
    s.close()


def test_self_matrix_batch(G):
    rng = np.random.default_rng(9)
    n, nf = 700, 5
    s = G.System(n, n_slots=nf)
    boxes = [O.box_from_lengths_angles([5.0 + 0.1 * f, 5.5, 6.0], [80.0, 85.0 - f, 75.0]) for f in range(nf)]
    frames = []
    for f in range(nf):
        b = boxes[f]
        frames.append((rng.random((n, 3)) @ np.array([[b[0], 0, 0], [b[5], b[1], 0], [b[7], b[8], b[2]]], np.float64)).astype(np.float32))
        s.set_frame(frames[f], b, slot=f)
    s.group_create_from_ranges("S", [(0, n - 1)])
    res = {}
    for sym in (1, 0):
        s.set_tuning(pairdist_symmetric=sym)
        dev, n1, n2, status = s.group_all_distances_batch_device("S", "S", 0, nf, G.Dimension.XYZ)
        assert (n1, n2) == (n, n) and (np.asarray(status) == 0).all()
        res[sym] = s.device_read(dev, 0, (nf, n, n))
    assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))
    np.testing.assert_allclose(res[1][3], O.group_all_distances(frames[3], np.arange(n), np.arange(n), "xyz", boxes[3]), atol=2e-6, rtol=0)
    s.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_cells_and_sizes(G, seed):
    """random GROMACS-reduced cells (|bx| <= ax/2, |cx| <= ax/2, |cy| <= by/2; some flat, some on the limits), random sizes: the
    symmetric kernel gives the bits of the plain one in every Dimension, XYZ agrees with the oracle (whose image table is checked
    against an fp64 lattice search in the CPU suite); cells too skewed for the table are refused by both"""
    rng = np.random.default_rng(900 + seed)
    ax, by, cz = rng.uniform(3.0, 9.0, 3)
    if seed % 3 == 0:
        cz = rng.uniform(1.2, 2.5)                                   # flat
    f = rng.uniform(-0.5, 0.5, 3)
    if seed % 4 == 1:
        f = np.sign(f) * 0.5                                          # on the limits
    box = np.array([ax, by, cz, 0, 0, f[0] * ax, 0, f[1] * ax, f[2] * by], np.float32)
    L = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    n = int(rng.integers(256, 2200))
    pos = (rng.uniform(-0.2, 1.2, (n, 3)) @ L).astype(np.float32)
    s = G.System(n, n_slots=1)
    s.set_frame(pos, box, slot=0)
    s.group_create_from_ranges("S", [(0, n - 1)])
    try:
        a, b = both(G, s, "S", "XYZ")
    except G.DeviceError as e:
        assert "skewed" in str(e)
        s.close()
        return
    idx = np.arange(n)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    want = O.group_all_distances(pos, idx, idx, "xyz", box)
    tol = 3e-6 + 2.5e-7 * float((box[:3].astype(np.float64) ** 2).sum()) / np.maximum(want, 1e-3)   # (length-only search: its stated cancellation error)
    assert (np.abs(a - want) <= tol).all(), float(np.abs(a - want).max())
    for dim in ("X", "Z", "XY", "YZ"):
        a, b = both(G, s, "S", dim)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), dim
        assert np.array_equal(a, -a.T) if len(dim) == 1 else np.array_equal(a, a.T), dim
    s.close()
