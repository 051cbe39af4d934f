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
