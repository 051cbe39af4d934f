"""trr frames straight into device slots (gr_trr_read_frames_device: raw big-endian positions over PCIe, byte order and
precision converted on the GPU) must equal the host reader (gr_trr_read_frame), which equals the reference's reader
(tests/test_trr_reader.py); all-zero positions arrive as the missing-position marker (trr_io.rs:108-112)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


@pytest.mark.parametrize("name", ["triclinic_trajectory.trr", "triclinic_trajectory_double_precision.trr", "dodecahedron_trajectory_full.trr",
                                  "octahedron_trajectory.trr", "short_trajectory_protein.trr"])
def test_device_conversion_equals_the_host_reader(G, name):
    t = G.TrrFile(os.path.join(GOLD, name))
    nb = 4
    s = G.System(t.n_atoms, n_slots=nb)
    for f0 in range(0, t.n_frames, nb):
        n = min(nb, t.n_frames - f0)
        steps, times = t.read_frames_device(s, f0, n)
        for k in range(n):
            x, _, _, box9, step, time, _ = t.read_frame(f0 + k)
            got = s.get_positions(k)
            zero = ~x.any(axis=1)
            assert np.array_equal(np.isnan(got[:, 0]), zero)
            assert np.array_equal(got[~zero], x[~zero])
            assert steps[k] == step and times[k] == np.float32(time)
            if box9 is not None:
                assert np.array_equal(s.get_box(k), box9)
    # strided frames, then an analysis on what arrived
    steps, _ = t.read_frames_device(s, 0, 3, frame_step=2)
    assert [int(v) for v in steps] == [t.frame_info(k)["step"] for k in (0, 2, 4)]
    if t.frame_info(0)["positions"]:
        np.testing.assert_allclose(s.group_get_center_naive("all", slot=0), t.read_frame(0)[0].mean(0), atol=1e-5)
    s.close(); t.close()
