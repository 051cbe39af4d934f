"""GPU parity of the pair-distance kernel's specialised inner loops and of the batched call
(reference: System::group_all_distances, src/system/analysis.rs:401-427 over Vector3D::distance, src/structures/vector3d.rs:458-486,
min_image :575-592).  The orthorhombic "near" loop must equal the reference's while-loops bit for bit in the signed 1-D
dimensions (one f32 subtraction either way); atoms far outside the cell take the generic closed form."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
DIMS = ["X", "Y", "Z", "XY", "XZ", "YZ", "XYZ"]


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def make(G, pos, box, n_slots=1):
    s = G.System(pos.shape[0], n_slots=n_slots)
    s.set_frame(pos, box, slot=0)
    s.group_create_from_ranges("A", [(0, 36)])
    s.group_create_from_ranges("B", [(30, pos.shape[0] - 1)])
    return s


def check(G, s, pos, box, slot=0, exact_1d=True):
    ia, ib = np.arange(0, 37), np.arange(30, pos.shape[0])
    for dim in DIMS:
        got = s.group_all_distances("A", "B", G.Dimension[dim], slot=slot)
        want = O.group_all_distances(pos, ia, ib, dim.lower(), box)
        if len(dim) == 1 and exact_1d:
            assert np.array_equal(got, want), dim
        else:
            np.testing.assert_allclose(got, want, atol=2e-6, rtol=0, err_msg=dim)


@pytest.mark.parametrize("n", [1061, 4096, 5000])
def test_orthorhombic_near_loop_bit_exact_in_1d(G, n):
    """every atom inside the cell (plus values exactly on 0, L, and half-box separations): the near loop"""
    rng = np.random.default_rng(n)
    box = np.array([6.5, 7.25, 5.0, 0, 0, 0, 0, 0, 0], np.float32)
    pos = (rng.random((n, 3)) * box[:3]).astype(np.float32)
    pos[0] = [0.0, 0.0, 0.0]; pos[31] = box[:3]; pos[32] = box[:3] / 2            # separations of exactly L and L/2
    pos[40] = [3.25, 3.625, 2.5]; pos[41] = [np.nextafter(np.float32(3.25), np.float32(9)), 3.625, np.nextafter(np.float32(2.5), np.float32(0))]
    pos[1] = -0.24 * box[:3]; pos[50] = 1.24 * box[:3]                               # the edge of the near licence: |d| ~ 1.48 L
    s = make(G, pos, box)
    check(G, s, pos, box)
    s.close()


def test_orthorhombic_far_atoms_take_the_generic_path(G):
    """atoms several boxes away in some tiles only: those tiles use the closed form, the others the near loop"""
    rng = np.random.default_rng(3)
    n = 6000
    box = np.array([4.0, 5.0, 6.0, 0, 0, 0, 0, 0, 0], np.float32)
    pos = (rng.random((n, 3)) * box[:3]).astype(np.float32)
    pos[5] += np.float32([8.0, -10.0, 12.0]); pos[2000] -= np.float32([4.0 * 3, 0, 6.0 * 2]); pos[5999, 1] = 27.3
    s = make(G, pos, box)
    check(G, s, pos, box, exact_1d=False)    # |k| >= 2: one rounding here, repeated subtraction in the reference (an ulp)
    s.close()


@pytest.mark.parametrize("angles", [[60.0, 60.0, 90.0], [70.53, 109.47, 70.53], [75.0, 80.0, 70.0]])
def test_triclinic_packed_loop_vs_oracle(G, angles):
    rng = np.random.default_rng(11)
    n = 3000
    box = O.box_from_lengths_angles([5.0, 5.0, 5.0] if angles[0] != 75.0 else [6.0, 5.5, 5.0], angles)
    frac = rng.random((n, 3))
    pos = (frac[:, :1] * [box[0], 0, 0] + frac[:, 1:2] * [box[5], box[1], 0] + frac[:, 2:] * [box[7], box[8], box[2]]).astype(np.float32)
    pos[7] += np.float32([box[0] * 2, 0, 0])         # one atom two cells away along a
    s = make(G, pos, box)
    ia, ib = np.arange(0, 37), np.arange(30, n)
    got = s.group_all_distances("A", "B", G.Dimension.XYZ)
    want = O.group_all_distances(pos, ia, ib, "xyz", box)
    np.testing.assert_allclose(got, want, atol=1e-5, rtol=0)
    # 1-D (signed) and 2-D dimensions: components of the 3-D minimum-image vector (vector3d.rs:458-486 generalised) -- the packed
    # search that names the winning image.  A pair whose two best images are equally long to within the f32 rounding of their
    # squared lengths (~1e-7 relative) may legitimately resolve to either: such pairs are excluded, everything else must agree.
    L = np.array([[box[0], 0, 0], [box[5], box[1], 0], [box[7], box[8], box[2]]], np.float64)
    ks = np.array([(i, j, k) for i in range(-2, 3) for j in range(-2, 3) for k in range(-2, 3)], np.float64) @ L
    d = pos[ia].astype(np.float64)[:, None, :] - pos[ib].astype(np.float64)[None, :, :]
    r2 = np.sort(((d[:, :, None, :] + ks[None, None, :, :]) ** 2).sum(-1), axis=-1)
    clear = (r2[:, :, 1] - r2[:, :, 0]) > 1e-4                                        # nm^2 between the best and the second-best image
    assert clear.mean() > 0.999
    for dim in ("X", "Y", "Z", "XY", "XZ", "YZ"):
        got = s.group_all_distances("A", "B", G.Dimension[dim])
        want = O.group_all_distances(pos, ia, ib, dim.lower(), box)
        assert np.abs(got - want)[clear].max() <= 1e-5, dim
    s.close()


def test_batch_matches_single_calls_and_reports_per_frame(G):
    rng = np.random.default_rng(21)
    n, nf = 2500, 5
    boxes = [np.array([5.0 + 0.1 * f, 6.0, 7.0 - 0.2 * f, 0, 0, 0, 0, 0, 0], np.float32) for f in range(nf)]
    boxes[3] = O.box_from_lengths_angles([5.0, 6.0, 7.0], [80.0, 85.0, 75.0])     # a triclinic frame among orthorhombic ones
    s = G.System(n, n_slots=nf)
    frames = []
    for f in range(nf):
        pos = (rng.random((n, 3)) * boxes[f][:3]).astype(np.float32)
        frames.append(pos); s.set_frame(pos, boxes[f], slot=f)
    s.group_create_from_ranges("A", [(0, 36)])
    s.group_create_from_ranges("B", [(30, n - 1)])
    for dim in ("XYZ", "Y", "XZ"):
        dev, n1, n2, status = s.group_all_distances_batch_device("A", "B", 0, nf, G.Dimension[dim])
        assert (n1, n2) == (37, n - 30) and (status == 0).all()
        for f in range(nf):
            got = s.device_read(dev, f * n1 * n2, (n1, n2))
            assert np.array_equal(got, s.group_all_distances("A", "B", G.Dimension[dim], slot=f)), (dim, f)
    bad = frames[2].copy(); bad[33, 0] = np.nan        # in both groups: the row atom is met first (analysis.rs:414-424)
    s.set_frame(bad, boxes[2], slot=2)
    dev, n1, n2, status = s.group_all_distances_batch_device("A", "B", 0, nf, raise_on_error=False)
    assert status[2] != 0 and all(status[f] == 0 for f in (0, 1, 3, 4))
    with pytest.raises(G.GroupError) as e:
        s.group_all_distances_batch_device("A", "B", 0, nf)
    assert e.value.variant == "InvalidPosition" and e.value.detail == 33
    got = s.device_read(dev, 4 * n1 * n2, (n1, n2))
    np.testing.assert_allclose(got, O.group_all_distances(frames[4], np.arange(37), np.arange(30, n), "xyz", boxes[4]), atol=2e-6, rtol=0)
    s.close()
