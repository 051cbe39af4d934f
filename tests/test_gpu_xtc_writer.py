"""Fitted-trajectory output on the GPU box: xtc decode -> RMSD-fit on the device -> gr_xtc_write_slots (D2H + the library's
encoder) against the reference's golden fitted trajectory (short_trajectory_fit.xtc, src/system/rmsd.rs:950-1073; the kept
atoms are in tests/golden/short_traj.npz) and, byte for byte, against the reference's own writer on the same coordinates."""
import os

import numpy as np
import pytest

from test_xtc_decoder import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def test_decode_fit_write_matches_the_reference_fit_golden(G, tmp_path, example, short_traj):
    x = G.XtcFile(os.path.join(GOLD, "short_trajectory.xtc"))
    n, nf = x.n_atoms, x.n_frames
    masses = np.full(n, np.nan, np.float32); masses[:61] = example["protein_masses"]
    ref = G.System(n, masses=masses, box=example["box9"], positions=example["pos"])
    cur = G.System(n, masses=masses, n_slots=nf)
    for s in (ref, cur):
        s.group_create_from_ranges("Protein", [(0, 60)])
    steps, times = x.read_frames_device(cur, 0, nf)
    plan = G.RMSDPlan(ref, cur, "Protein")
    r, st = plan.rmsd_fit(0, nf)
    assert (st == 0).all()
    prec = x.frame_info(0)[3]
    out = tmp_path / "fit.xtc"
    with G.XtcWriter(out) as w:
        w.write_slots(cur, 0, nf, steps=steps.astype(np.int64), times=times, precision=prec)
    y = G.XtcFile(out)
    assert y.n_atoms == n and y.n_frames == nf
    keep = short_traj["keep"].astype(np.int64)
    # both files hold QUANTISED coordinates whose sources agree to ~1e-4 nm (the reference's own f32 SVD noise, see
    # test_oracle_golden.py): almost every coordinate is identical, a source that sits on a rounding boundary lands one
    # quantum apart
    flips = total = 0
    for f in range(nf):
        pos, box, step, time, p = y.read_frame(f)
        assert step == steps[f] and time == times[f] and p == np.float32(prec)
        assert np.array_equal(box, x.frame_info(f)[2])
        d = np.abs(pos[keep] - short_traj["fit"][f])
        assert d.max() <= 1.0 / prec + 1e-6
        flips += int((d > 1e-6).sum()); total += d.size
        # what was written is the quantised device frame
        assert np.abs(pos - cur.get_positions(f)).max() <= 0.5 / prec + 1e-6
    assert flips <= 0.03 * total, (flips, total)
    # the same frames through the host entry point: identical bytes
    out2 = tmp_path / "fit_host.xtc"
    with G.XtcWriter(out2) as w:
        for f in range(nf):
            w.write_frame(cur.get_positions(f), cur.get_box(f), step=int(steps[f]), time=float(times[f]), precision=prec)
    assert open(out, "rb").read() == open(out2, "rb").read()
    # a group writer (xtc_group_writer_init): only the group's atoms, in its order
    out3 = tmp_path / "fit_protein.xtc"
    with G.XtcWriter(out3) as w:
        w.write_slots(cur, 0, nf, group="Protein", steps=steps.astype(np.int64), times=times, precision=prec, host_threads=3)
    z = G.XtcFile(out3)
    assert z.n_atoms == 61 and z.n_frames == nf
    for f in (0, nf - 1):
        assert np.array_equal(z.read_frame(f)[0], y.read_frame(f)[0][:61])
    with pytest.raises(G.XtcError):
        G.XtcWriter(tmp_path / "x.xtc").write_slots(cur, 0, 1, group="nope")
    # (the encoder against the reference's writer byte for byte: tests/test_xtc_writer.py in the CPU suite, and the pinned files of
    # tests/test_gpu_xtc_device.py here)
    x.close(); y.close(); z.close(); ref.close(); cur.close()


def test_a_frame_the_format_cannot_hold_stops_the_output_there(tmp_path):
    """write_slots over four slots, the third holds a coordinate whose quantum does not fit 32 bits: GR_E_OUT_OF_RANGE, and the
    file holds the two frames before it -- what a loop of write_frame calls would leave (the reference's C writer prints
    "Internal overflow compressing coordinates." and converts anyway: undefined behaviour, xdrfile.c:1025-1030)"""
    import groan_rs_amd as G
    rng = np.random.default_rng(8)
    n = 500
    box = np.array([5, 5, 5, 0, 0, 0, 0, 0, 0], np.float32)
    s = G.System(n, n_slots=4)
    frames = [rng.uniform(0, 5, (n, 3)).astype(np.float32) for _ in range(4)]
    frames[2][77, 1] = 3.0e6
    for f in range(4):
        s.set_frame(frames[f], box, slot=f)
    path = tmp_path / "stop.xtc"
    with G.XtcWriter(path) as w:
        with pytest.raises(G.XtcError) as e:
            w.write_slots(s, 0, 4, host_threads=3)
        assert e.value.status == 9                                      # GR_E_OUT_OF_RANGE
        w.write_slots(s, 3, 1)                                          # the writer is still usable
    x = G.XtcFile(path)
    assert x.n_frames == 3
    for k, f in enumerate((0, 1, 3)):
        assert np.abs(x.read_frame(k)[0] - frames[f]).max() <= 0.00051
    x.close(); s.close()
