"""BASELINE.json configs[0] and configs[1] as stated, against committed golden values AND the oracle:

  C1  example.gro + short_trajectory.xtc: per-frame centre of mass of the Protein group -- frames
      decoded by the library's own xtc reader (host and device paths), COM by gr_group_center(_batch)
  C2  aa_membrane_peptide.xtc: RMSD of the peptide to the first frame, orthorhombic PBC, per frame

Golden values: tests/golden/c1_c2_expected.json, written by tests/golden/make_c1_c2_golden.py with the oracle in its literal
f32 mode (the oracle itself is pinned by the reference's known answers, tests/test_oracle_golden.py).  Reference entry points:
System::group_get_com (analysis.rs:258-274), System::calc_rmsd / RMSDTrajRead::calc_rmsd (rmsd.rs:75-166,258-401)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


def expected():
    with open(os.path.join(GOLD, "c1_c2_expected.json")) as fh:
        return json.load(fh)


def test_oracle_reproduces_the_committed_c1_c2_values(short_traj, example, aa):
    """(CPU) the committed goldens are what the pinned oracle computes from the committed fixtures -- a change in the oracle
    that moves them is caught here, not silently absorbed by regenerating the file"""
    e = expected()
    m = example["protein_masses"]
    for f in range(short_traj["frames"].shape[0]):
        got = O.get_center(short_traj["frames"][f][:61], np.arange(61), short_traj["boxes9"][f], mass=m)
        assert np.array_equal(got, np.array(e["c1_protein_com"][f], np.float32))
    b = aa["blocks_peptide"][0]
    pm = aa["masses"][int(b[0]):int(b[1]) + 1]
    tr, tb = aa["traj_peptide"], aa["traj_boxes9"]
    for f in range(tr.shape[0]):
        r, _ = O.calc_rmsd(tr[0], pm, np.arange(tr.shape[1]), tb[0], tr[f], pm, np.arange(tr.shape[1]), tb[f])
        assert np.float32(r) == np.float32(e["c2_peptide_rmsd_to_first_frame"][f])
    # the first frame of short_trajectory.xtc is not example.gro, but the pinned COM of example.gro's Protein
    # (analysis.rs:813-989: 4.485 / 3.188 / 1.73549 is another group) keeps the masses honest: sum = 3204
    assert abs(float(m.sum()) - 3204.0) < 1e-3


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


@pytest.mark.gpu
@pytest.mark.parametrize("device_decode", [False, True])
def test_config1_per_frame_protein_com(G, example, device_decode):
    e = expected()
    x = G.XtcFile(os.path.join(GOLD, "short_trajectory.xtc"))
    n, nf = x.n_atoms, x.n_frames
    assert (n, nf) == (16844, 11)
    masses = np.full(n, np.nan, np.float32)          # only the Protein beads carry masses in the fixture (example.tpr)
    masses[:61] = example["protein_masses"]
    s = G.System(n, masses=masses, n_slots=nf)
    s.group_create_from_ranges("Protein", [(0, 60)])
    host = [x.read_frame(f) for f in range(nf)]
    if device_decode:
        x.read_frames_device(s, 0, nf)
    else:
        for f in range(nf):
            s.set_frame(host[f][0], host[f][1], slot=f)
    batch, bst = s.group_get_com_batch("Protein", 0, nf)
    assert (bst == 0).all()
    idx = np.arange(61)
    for f in range(nf):
        got = s.group_get_com("Protein", slot=f)
        assert np.array_equal(got, batch[f])
        want = O.get_center(host[f][0], idx, host[f][1], mass=masses)                 # the oracle on the full decoded frame
        assert np.abs(got - want).max() <= TOL, (f, got, want)
        assert np.abs(got - np.array(e["c1_protein_com"][f], np.float32)).max() <= TOL
        gc = s.group_get_center("Protein", slot=f)
        assert np.abs(gc - np.array(e["c1_protein_center"][f], np.float32)).max() <= TOL
    s.close(); x.close()


@pytest.mark.gpu
def test_config2_peptide_rmsd_to_first_frame(G, aa):
    e = expected()
    b = aa["blocks_peptide"][0]
    tr, tb = aa["traj_peptide"], aa["traj_boxes9"]
    nf, n = tr.shape[0], tr.shape[1]
    pm = aa["masses"][int(b[0]):int(b[1]) + 1]
    ref = G.System(n, masses=pm, box=tb[0], positions=tr[0])
    cur = G.System(n, masses=pm, n_slots=nf)
    for f in range(nf):
        cur.set_frame(tr[f], tb[f], slot=f)
    plan = G.RMSDPlan(ref, cur, "all")
    for exact in (False, True):
        plan.force_exact(exact)
        r, st = plan.rmsd(0, nf)
        assert (st == 0).all()
        assert np.abs(r - np.array(e["c2_peptide_rmsd_to_first_frame"], np.float32)).max() <= TOL
    # the per-frame call a trajectory loop makes (System::calc_rmsd)
    for f in (0, 7, 20):
        assert abs(cur.calc_rmsd(ref, "all", slot=f) - e["c2_peptide_rmsd_to_first_frame"][f]) <= TOL
    plan.close(); ref.close(); cur.close()


@pytest.mark.gpu
def test_config2_as_a_selection_inside_the_full_system(G, aa):
    """configs[1] at its real shape: the 32 817-atom membrane system, `@protein` chosen by the selection language
    (groups.rs:36-92, select/mod.rs) out of the structure's residue names, frames 0 and 20 of aa_membrane_peptide.xtc (fixture
    aa_full.npz, decoded with the reference's vendored xdrfile: tests/golden/make_c2_full_fixture.py).  System::calc_rmsd and
    calc_rmsd_and_fit (rmsd.rs:75-166; the iterator form :1202-1226) against the committed golden RMSD of frame 20, the oracle on the
    full system, and -- for the fit -- every one of the 32 817 atoms (rmsd.rs:508-528 moves the whole system, not the selection)."""
    import types
    from groan_rs_amd.select import group_create
    e = expected()
    d = np.load(os.path.join(GOLD, "aa_full.npz"))
    frames, boxes = d["frames"], d["boxes9"]
    n = frames.shape[1]
    assert n == 32817 and list(d["frame_index"]) == [0, 20]
    structure = types.SimpleNamespace(n_atoms=n, resid=d["resid"], atomid=d["atomid"], resname=[x.decode() for x in d["resname"]], atomname=[x.decode() for x in d["atomname"]])
    masses = aa["masses"]
    ref = G.System(n, masses=masses, box=boxes[0], positions=frames[0])
    cur = G.System(n, masses=masses, n_slots=2)
    for f in range(2):
        cur.set_frame(frames[f], boxes[f], slot=f)
    for s in (ref, cur):
        group_create(s, "Protein", "@protein", structure)
    idx = np.array(list(cur.group_container("Protein")), np.int64)
    assert idx.size == 363 and np.array_equal(idx, np.arange(363))                      # bit-exact selection (indices 0-362)
    # the same coordinates as the peptide-only fixture the golden values were computed from
    assert np.array_equal(frames[0][:363], aa["traj_peptide"][0]) and np.array_equal(frames[1][:363], aa["traj_peptide"][20])
    want = [e["c2_peptide_rmsd_to_first_frame"][0], e["c2_peptide_rmsd_to_first_frame"][20]]
    for small in (4096, 0):                                                            # the single-wave call and the batched kernels
        cur.set_tuning(small_calls=small)
        for f in range(2):
            assert abs(cur.calc_rmsd(ref, "Protein", slot=f) - want[f]) <= TOL, (small, f)
        plan = G.RMSDPlan(ref, cur, "Protein")
        r, st = plan.rmsd(0, 2)
        assert (st == 0).all() and np.abs(r - np.array(want, np.float32)).max() <= TOL
        plan.close()
    # the oracle inside the full system, and the fit of ALL atoms
    for small in (4096, 0):
        cur.set_tuning(small_calls=small)
        cur.set_frame(frames[1], boxes[1], slot=1)
        ro, fitted = O.calc_rmsd_and_fit(frames[0], masses, idx, boxes[0], frames[1], masses, idx, boxes[1])
        assert abs(ro - want[1]) <= 2e-6
        got = cur.calc_rmsd_and_fit(ref, "Protein", slot=1)
        assert abs(got - want[1]) <= TOL
        out = cur.get_positions(1)
        assert out.shape == (n, 3) and np.abs(out - fitted).max() <= 5e-5, (small, float(np.abs(out - fitted).max()))
        assert np.abs(out[363:] - frames[1][363:]).max() > 1e-3                          # (the membrane and the water moved with the peptide)
    ref.close(); cur.close()
