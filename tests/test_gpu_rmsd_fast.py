"""calc_rmsd (RMSD WITHOUT fit, rmsd.rs:75-129,141-166) through the one read-only pass k_sums_pk<false, true>: the fit path's sums +
the closed-form RMSD's sums as short f32 chains widened to fp64 (gr_hot.h), with the closing step's own estimate of the rounding
left in rmsd^2 deciding which frames go back to the exact-product pass (GR_ST_REDO_EXACT).  Checked here: the oracle at full size;
the exact pass on the same frames (the estimate must bound the observed difference by a wide margin at every size); rigid copies
and near-copies of the reference (rmsd ~ 0: handed back, and still right); the pinned trajectory RMSDs of the reference with the
pass forced onto a 61-atom group; frames whose image proof fails; selections, weights and shapes the pass must refuse."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from groan_rs_amd import workload as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def _blob(G, n, nf, box, noise, sel=None):
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, noise, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    name = "all"
    if sel is not None:
        name = "S"
        for s in (ref, cur):
            s.group_create_from_ranges("S", [sel])
    return masses, cur, ref, ref_pos, G.RMSDPlan(ref, cur, name)


def test_full_size_against_the_oracle_and_the_exact_pass(G):
    """1e6 atoms, rhombic dodecahedron (BASELINE configs[3]'s frames), 24 frames: the f32-chain pass closes every frame, agrees with
    the oracle (sums in double) to 1e-5 nm and with the exact-product pass to a small fraction of that; rotations to 2e-6"""
    n, nf = 1_000_000, 24
    box = W.c4_box()
    masses, cur, ref, ref_pos, plan = _blob(G, n, nf, box, 0.05)
    cur.profile_enable(True)
    r, st, R = plan.rmsd(0, nf, return_rotation=True)
    prof = cur.profile_read()
    assert (st == 0).all() and plan.last_fallbacks() == 0
    assert cur.stat("rmsd_fast_frames") == nf and cur.stat("rmsd_exact_redos") == 0
    assert prof["k_sums_pk"][1] == 1 and prof["k_fit_pk"][1] == 0 and prof["k_fit_resident"][1] == 0, prof
    cur.set_tuning(rmsd_fast=0)
    r0, st0, R0 = plan.rmsd(0, nf, return_rotation=True)
    assert cur.stat("rmsd_fast_frames") == nf                                   # (the exact pass does not count)
    assert np.abs(np.asarray(r) - np.asarray(r0)).max() <= 5e-7, np.abs(np.asarray(r) - np.asarray(r0)).max()
    assert np.abs(R - R0).max() <= 2e-6
    idx = np.arange(n)
    with O.acc64():
        for f in (0, 11, nf - 1):
            ro = O.calc_rmsd(ref_pos, masses, idx, box, cur.get_positions(f), masses, idx, box)[0]
            assert abs(float(r[f]) - ro) <= 1e-5, (f, float(r[f]), ro)
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("n,cell,noise", [(20_000, "ortho", 0.05), (20_000, "tric", 0.2), (123_457, "dodeca", 0.03), (400_000, "tric", 0.08)])
def test_the_error_estimate_bounds_what_the_chains_lose(G, n, cell, noise):
    """sizes from 2e4 to 4e5 atoms (the pass forced below its default 16 384-atom threshold is covered further down), three cells, a
    selection with ragged ends: |fast - exact| stays below 1e-6 nm wherever the pass keeps a frame"""
    nf = 12
    box = {"ortho": W.box_from_lengths_angles([9.0, 8.0, 7.0], [90.0, 90.0, 90.0]), "tric": W.box_from_lengths_angles([9.0, 8.5, 8.0], [75.0, 80.0, 70.0]),
           "dodeca": W.c4_box(9.0)}[cell]
    masses, cur, ref, ref_pos, plan = _blob(G, n, nf, box, noise, sel=(7, n - 6))
    r, st = plan.rmsd(0, nf)
    kept = cur.stat("rmsd_fast_frames")
    assert (st == 0).all() and kept + cur.stat("rmsd_exact_redos") == nf and kept >= nf // 2, (kept, cur.stat("rmsd_exact_redos"))
    cur.set_tuning(rmsd_fast=0)
    r0, _ = plan.rmsd(0, nf)
    assert np.abs(np.asarray(r) - np.asarray(r0)).max() <= 1e-6, np.abs(np.asarray(r) - np.asarray(r0)).max()
    idx = np.arange(7, n - 5)
    with O.acc64():
        ro = O.calc_rmsd(ref_pos, masses, idx, box, cur.get_positions(3), masses, idx, box)[0]
    assert abs(float(r[3]) - ro) <= 1e-5
    plan.close(); ref.close(); cur.close()


def test_rigid_and_nearly_rigid_copies_are_handed_back_and_still_right(G):
    """rmsd^2 is a small difference of sums of size W r^2: for a rigid copy of the reference (rmsd = 0; the reference's tests ask
    |rmsd| <= 1e-4 there, rmsd.rs:618-780) and for copies with 1e-3 / 5e-3 nm of noise the f32 chains cannot say; the closing step
    must notice and the exact-product pass must answer.  Noise of 0.05 nm is kept."""
    n = 300_000
    box = W.c4_box(16.0)
    masses = W.masses_cycle(n)
    noises = [0.0, 0.0, 1e-3, 5e-3, 0.05, 0.0, 0.05, 0.05]
    nf = len(noises)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    for f, s in enumerate(noises):
        cur.synth_frames(nf, f, 1, f, s, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    plan = G.RMSDPlan(ref, cur, "all")
    r, st = plan.rmsd(0, nf)
    assert (st == 0).all()
    redone = cur.stat("rmsd_exact_redos")
    assert redone >= 4 and cur.stat("rmsd_fast_frames") == nf - redone and cur.stat("rmsd_fast_frames") >= 3, (redone, cur.stat("rmsd_fast_frames"))
    idx = np.arange(n)
    with O.acc64():
        for f in range(nf):
            ro = O.calc_rmsd(ref_pos, masses, idx, box, cur.get_positions(f), masses, idx, box)[0]
            assert abs(float(r[f]) - ro) <= 1e-5, (f, noises[f], float(r[f]), ro)
            if noises[f] == 0.0:
                assert float(r[f]) <= 1e-4                                       # the reference's own bar for a rigid copy
    plan.close(); ref.close(); cur.close()


def test_pinned_trajectory_rmsds_with_the_pass_forced_onto_a_small_group(G):
    """the reference's 11 pinned RMSDs (rmsd.rs:811-814: Protein(61) of example.tpr against short_trajectory.xtc) with the threshold
    lowered to zero: 61 atoms in one chunk -- whatever the closing step keeps must agree with the pinned values like the exact pass
    does (<= 5e-7), whatever it hands back likewise"""
    g = np.load(os.path.join(HERE, "golden", "example.npz"))
    t = np.load(os.path.join(HERE, "golden", "short_traj.npz"))
    want = [0.23669721, 0.2634763, 0.26021627, 0.21364464, 0.22166993, 0.19383307, 0.26422343, 0.27013618, 0.26398134, 0.23475659, 0.24208021]
    keep = t["keep"].astype(np.int64)
    masses = np.full(keep.size, np.nan, np.float32); masses[:61] = g["protein_masses"]
    ref = G.System(keep.size, masses=masses, box=g["box9"], positions=t["gro_keep"])
    cur = G.System(keep.size, masses=masses, n_slots=11)
    for s in (ref, cur):
        s.group_create_from_ranges("Protein", [(0, 60)])
    for f in range(11):
        cur.set_frame(t["frames"][f], t["boxes9"][f], slot=f)
    cur.set_tuning(rmsd_fast_min=0)
    plan = G.RMSDPlan(ref, cur, "Protein")
    r, st = plan.rmsd(0, 11)
    assert (st == 0).all() and cur.stat("rmsd_fast_frames") + cur.stat("rmsd_exact_redos") == 11
    assert np.abs(np.asarray(r) - np.asarray(want, np.float32)).max() <= 5e-7, (r, cur.stat("rmsd_fast_frames"))
    plan.close(); ref.close(); cur.close()


def test_frames_whose_image_proof_fails_and_frames_without_a_position(G):
    """the pass shares the fit path's image proof: a group wider than half the cell goes to the literal multi-pass path, a frame with a
    missing position reports it, the frames around them are closed by the pass"""
    n, nf = 60_000, 9
    box = W.box_from_lengths_angles([9.0, 8.5, 8.0], [75.0, 80.0, 70.0])
    masses, cur, ref, ref_pos, plan = _blob(G, n, nf, box, 0.05)
    frames = [cur.get_positions(f) for f in range(nf)]
    frames[2] = W.proof_failing_frame(ref_pos, box, "two_lobes", 5)
    frames[6] = frames[6].copy(); frames[6][4242] = np.nan
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r, st = plan.rmsd(0, nf, raise_on_error=False)
    assert [f for f in range(nf) if st[f] != 0] == [6] and plan.last_fallbacks() >= 1
    idx = np.arange(n)
    with O.acc64():
        for f in (1, 2, 3, 8):
            ro = O.calc_rmsd(ref_pos, masses, idx, box, frames[f], masses, idx, box)[0]
            assert abs(float(r[f]) - ro) <= 1e-5, (f, float(r[f]), ro)
    cur.set_tuning(rmsd_fast=0)
    r0, st0 = plan.rmsd(0, nf, raise_on_error=False)
    assert np.array_equal(st, st0)
    plan.close(); ref.close(); cur.close()


def test_what_the_pass_must_refuse(G):
    """weights that are not the target's masses, scattered selections, groups below the threshold: the exact-product pass, as before"""
    n, nf = 40_000, 4
    box = W.box_from_lengths_angles([8.0, 8.0, 8.0], [90.0, 90.0, 90.0])
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    ref_pos = cur.get_positions(nf)
    other = masses[::-1].copy()
    ref = G.System(n, masses=other, box=box, positions=ref_pos)                   # reference masses (= weights) differ from the target's
    for s in (ref, cur):
        s.group_create_from_ranges("Two", [(0, 19_999), (20_010, n - 1)])
        s.group_create_from_ranges("Small", [(0, 9_999)])
    idx = np.arange(n)
    for name, sel_idx in (("all", idx), ("Two", np.r_[0:20_000, 20_010:n]), ("Small", np.arange(10_000))):
        plan = G.RMSDPlan(ref, cur, name)
        r, st = plan.rmsd(0, nf)
        assert (st == 0).all() and cur.stat("rmsd_fast_frames") == 0 and cur.stat("rmsd_exact_redos") == 0, name
        with O.acc64():
            ro = O.calc_rmsd(ref_pos, other, sel_idx, box, cur.get_positions(1), masses, sel_idx, box)[0]
        assert abs(float(r[1]) - ro) <= 1e-5, (name, float(r[1]), ro)
        plan.close()
    ref.close(); cur.close()


def _lattice(n, spacing, centre):
    """n points of a simple-cubic lattice (spacing nm) filling a cube about `centre`: identical fractional parts along every row -- the
    opposite of the Gaussian blobs the pass's error estimate was calibrated on"""
    k = int(np.ceil(n ** (1.0 / 3.0)))
    g = np.stack(np.meshgrid(np.arange(k), np.arange(k), np.arange(k), indexing="ij"), -1).reshape(-1, 3)[:n].astype(np.float64)
    return ((g - (k - 1) / 2.0) * spacing + np.asarray(centre, np.float64))


def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)], [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


@pytest.mark.parametrize("n", [20_000, 1_000_000])
def test_hostile_inputs_lattices_and_quantised_coordinates(G, n):
    """The pass keeps a frame on a STATISTICAL model of what its f32 chains lose (independent roundings, a random walk: gr_kernels.h,
    gr_finalize_math<.., FAST>), calibrated on Gaussian-noise blobs.  Inputs whose roundings are NOT independent:
      * every coordinate a multiple of 0.001 nm -- what every xtc file delivers (precision 1000);
      * a crystalline reference: a simple-cubic lattice, thousands of atoms sharing each coordinate value;
      * frames that are that lattice rotated and displaced (large products, structured), the lattice plus 1e-4 nm of noise (rmsd ~ 1e-4:
        the guard must hand it to the exact pass or be right), and the lattice itself (rmsd 0).
    For every frame: |fast - exact| <= 2.5e-6 nm -- the guard a kept frame was kept under -- and <= 1e-5 nm against the oracle (fp64 sums)."""
    rng = np.random.default_rng(n)
    spacing = 0.300 if n <= 20_000 else 0.100
    L = 40.0
    box = W.box_from_lengths_angles([L, L, L], [90.0, 90.0, 90.0])
    centre = np.array([L / 2, L / 2, L / 2])
    refx = np.round(_lattice(n, spacing, centre) * 1000.0) / 1000.0
    masses = W.masses_cycle(n)
    frames = []
    # 0: the lattice itself; 1: + 1e-4 nm noise; 2: + 0.02 nm noise, quantised; 3-4: rotated about its centre and displaced, quantised; 5: a
    # PBC-broken copy (displaced across the cell faces and wrapped), quantised
    frames.append(refx.copy())
    frames.append(refx + rng.normal(0, 1e-4, refx.shape))
    frames.append(np.round((refx + rng.normal(0, 0.02, refx.shape)) * 1000.0) / 1000.0)
    for _ in range(2):
        R = _rot(rng)
        frames.append(np.round((((refx - centre) @ R.T) + centre + rng.uniform(-3, 3, 3) + rng.normal(0, 0.01, refx.shape)) * 1000.0) / 1000.0)
    moved = np.round((refx + np.array([L / 2 - 1.0, 3.0, -L / 2 + 2.0]) + rng.normal(0, 0.03, refx.shape)) * 1000.0) / 1000.0
    frames.append(np.mod(moved, L))
    nf = len(frames)
    frames = [f.astype(np.float32) for f in frames]
    cur = G.System(n, masses=masses, n_slots=nf)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    ref = G.System(n, masses=masses, box=box, positions=refx.astype(np.float32))
    plan = G.RMSDPlan(ref, cur, "all")
    r, st = plan.rmsd(0, nf)
    kept, redone = cur.stat("rmsd_fast_frames"), cur.stat("rmsd_exact_redos")
    assert (st == 0).all() and kept + redone + plan.last_fallbacks() >= nf, (st, kept, redone)
    cur.set_tuning(rmsd_fast=0)
    r0, st0 = plan.rmsd(0, nf)
    assert (st0 == 0).all()
    d = np.abs(np.asarray(r, np.float64) - np.asarray(r0, np.float64))
    assert d.max() <= 2.5e-6, (d, r, r0, kept, redone)
    idx = np.arange(n)
    refpos32 = refx.astype(np.float32)
    with O.acc64():
        for f in range(nf):
            ro = O.calc_rmsd(refpos32, masses, idx, box, frames[f], masses, idx, box)[0]
            assert abs(float(r[f]) - ro) <= 1e-5 and abs(float(r0[f]) - ro) <= 1e-5, (f, float(r[f]), float(r0[f]), ro)
    # the tiny-rmsd frames are where the estimate matters: they were handed to the exact pass, or kept and right (asserted above)
    assert redone >= 1 or kept == nf
    plan.close(); ref.close(); cur.close()
