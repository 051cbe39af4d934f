"""Device-side xtc unpacking (gr_xtc_read_frames_device: host skims the framing, k_xtc_unpack decodes every 32-atom segment
of every frame of the batch in parallel) must be BIT-IDENTICAL to the host decoder (gr_xtc_read_frame), which is
bit-identical to the reference's decoders (tests/test_xtc_decoder.py).  Reference: XtcReader / update_system
(src/io/xtc_io/mod.rs, molly_xtc.rs:268-308, xdrfile_xtc.rs:42-104)."""
import os

import numpy as np
import pytest

import xtc_cases as XC
from test_xtc_decoder import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def check_file(G, path, batch=None, frame_step=1, host_threads=0):
    x = G.XtcFile(path)
    idx = list(range(0, x.n_frames, frame_step))
    nb = len(idx) if batch is None else batch
    s = G.System(x.n_atoms, n_slots=nb)
    for b0 in range(0, len(idx), nb):
        part = idx[b0:b0 + nb]
        steps, times = x.read_frames_device(s, part[0], len(part), first_slot=0, frame_step=frame_step, host_threads=host_threads)
        for k, fr in enumerate(part):
            pos, box, step, time, _ = x.read_frame(fr)
            got = s.get_positions(k)
            assert np.array_equal(got.view(np.uint32), pos.view(np.uint32)), (path, fr, np.abs(got - pos).max())
            assert np.array_equal(s.get_box(k), box) and steps[k] == step and times[k] == np.float32(time)
    s.close(); x.close()


@pytest.mark.parametrize("name", ["triclinic_trajectory.xtc", "octahedron_trajectory.xtc", "dodecahedron_trajectory.xtc", "short_trajectory.xtc"])
def test_reference_data_files(G, name):
    check_file(G, os.path.join(GOLD, name))
    check_file(G, os.path.join(GOLD, name), batch=4, frame_step=2, host_threads=3)    # ragged last batch, strided frames, slot reuse


@pytest.mark.parametrize("case", XC.CASES)
def test_every_branch_of_the_format(G, tmp_path, case):
    """files written HERE by the library's own encoder; their sha256 is the one of the file the reference's writer produces from
    the same coordinates (tests/golden/xtc_pins.json, made in the build container by make_xtc_pins.py)"""
    frames, box, prec = XC.branch_case(case)
    path = tmp_path / (case + ".xtc")
    XC.write_own(G, path, frames, box, prec)
    assert XC.sha256_file(path) == XC.pin(case)["sha256"], case
    check_file(G, path)


def test_decoded_slots_feed_the_kernels(G, short_traj, example):
    """decode -> (no host copy) -> centre of geometry of the Protein per frame == the same through read_frame + set_frame"""
    x = G.XtcFile(os.path.join(GOLD, "short_trajectory.xtc"))
    s = G.System(x.n_atoms, n_slots=x.n_frames)
    s.group_create_from_ranges("Protein", [(0, 60)])
    x.read_frames_device(s, 0, x.n_frames)
    got = [s.group_get_center("Protein", slot=f) for f in range(x.n_frames)]           # ordered behind the unpack by the slot events
    t = G.System(x.n_atoms, n_slots=1)
    t.group_create_from_ranges("Protein", [(0, 60)])
    for f in range(x.n_frames):
        pos, box, _, _, _ = x.read_frame(f)
        t.set_frame(pos, box)
        assert np.array_equal(t.group_get_center("Protein"), got[f])
    s.close(); t.close(); x.close()


@pytest.mark.parametrize("seed", range(20))
def test_random_systems_unpack_like_the_host_decoder(G, tmp_path, seed):
    """randomised frames (atom counts around the 32-atom checkpoint spacing and the format's thresholds, molecules of 1-9
    atoms = runs of every length, coincident atoms, several precisions and magnitudes) written with the library's encoder
    (byte-identical to the reference's writer, tests/test_xtc_writer.py): device unpack == host decode, bit for bit"""
    rng = np.random.default_rng(52000 + seed)
    n = int(rng.choice([10, 31, 32, 33, 63, 64, 65, 100, 1000, 4097, 20011]))
    prec = float(rng.choice([1000.0, 100.0, 10000.0]))
    span = float(rng.choice([0.5, 3.0, 12.0, 80.0, 3000.0]))
    mol = int(rng.integers(1, 10))
    path = tmp_path / "rand.xtc"
    with G.XtcWriter(path) as w:
        for f in range(5):
            base = rng.uniform(-span if seed % 3 == 0 else 0.0, span, ((n + mol - 1) // mol, 3))
            x = np.repeat(base, mol, axis=0)[:n] + rng.normal(0, float(rng.choice([0.0, 0.002, 0.05, 0.3])), (n, 3))
            x[rng.integers(0, 6, n) == 1] = x[0]
            w.write_frame(x.astype(np.float32), [span, span, span, 0, 0, 0, 0, 0, 0], step=f, time=0.1 * f, precision=prec)
    check_file(G, path)
    check_file(G, path, batch=2, host_threads=2)


def test_group_limited_device_read(G, tmp_path):
    """GroupXtcReader on the device path (gr_xtc_read_frames_device_group): BASELINE configs[1]'s shape -- 363 peptide atoms in
    front of a 32 817-atom system -- and a scattered group.  The group's atoms are bit for bit the full decode's, every other
    atom of the slot is left as it was ("all other atoms are left unchanged", molly_xtc.rs:585-587), box / step / time are set."""
    rng = np.random.default_rng(11)
    n, nf = 32817, 6
    frames = [XC.water_like(rng, n, 7.0) for _ in range(nf)]
    path = tmp_path / "aa_shaped.xtc"
    box = [6.44, 6.76, 7.26, 0, 0, 0, 0, 0, 0]
    with G.XtcWriter(path) as w:
        for k, fr in enumerate(frames):
            w.write_frame(fr, box, step=100 * k, time=0.5 * k, precision=1000.0)
    x = G.XtcFile(path)
    s = G.System(n, n_slots=nf)
    s.group_create_from_ranges("Peptide", [(0, 362)])
    s.group_create_from_indices("Scattered", list(range(5, 20000, 7)) + list(range(100, 140)))
    s.group_create_from_indices("Empty", [])
    sentinel = np.full((n, 3), -7.25, np.float32)
    for group, members in (("Peptide", np.arange(363)), ("Scattered", np.unique(np.concatenate([np.arange(5, 20000, 7), np.arange(100, 140)]))), ("Empty", np.arange(0))):
        for f in range(nf):
            s.set_frame(sentinel, [1, 1, 1, 0, 0, 0, 0, 0, 0], slot=f)
        steps, times = x.read_frames_device(s, 0, nf, group=group, host_threads=3)
        inside = np.zeros(n, bool); inside[members] = True
        for f in range(nf):
            full, fbox, step, time, _ = x.read_frame(f)
            got = s.get_positions(f)
            assert np.array_equal(got[inside].view(np.uint32), full[inside].view(np.uint32)), (group, f)
            assert (got[~inside] == -7.25).all(), (group, f)
            assert np.array_equal(s.get_box(f), fbox) and steps[f] == step and times[f] == np.float32(time)
    with pytest.raises(G.XtcError):
        x.read_frames_device(s, 0, nf, group="NoSuchGroup")
    # and the consumer: RMSD of the peptide to the first frame from group-limited reads == from full reads
    m = np.ones(n, np.float32)
    s.set_masses(m)
    ref = G.System(n, masses=m, box=box, positions=x.read_frame(0)[0]); ref.group_create_from_ranges("Peptide", [(0, 362)])
    plan = G.RMSDPlan(ref, s, "Peptide")
    x.read_frames_device(s, 0, nf, group="Peptide")
    r_group, _ = plan.rmsd(0, nf)
    x.read_frames_device(s, 0, nf)
    r_full, _ = plan.rmsd(0, nf)
    assert np.array_equal(r_group, r_full)
    plan.close(); ref.close(); s.close(); x.close()
