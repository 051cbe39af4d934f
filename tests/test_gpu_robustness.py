"""Misuse the C ABI the way the round-1 review did: selections outside the system, a slot re-uploaded while its
batch is still in flight, groups / masses changed between gr_rmsd_batch_begin and _end.  None of it may reach a kernel with
bad indices or silently change results."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def test_selection_outside_the_system_never_becomes_a_group(G):
    n = 1000                                  # slots are padded to 1024 atoms
    s = G.System(n, box=[5, 5, 5], positions=np.random.default_rng(0).uniform(0, 5, (n, 3)).astype(np.float32))
    for bad, index in (([n], n), ([n, n + 1], n), ([1024 + 3], 1027), ([2 ** 40], 2 ** 40)):
        with pytest.raises(G.AtomError) as e:
            s.group_create_from_indices("Bad", bad)
        assert e.value.variant == "OutOfRange" and e.value.detail == index
        assert not s.group_exists("Bad")
    # an existing group survives a refused overwrite
    s.group_create_from_indices("Keep", [1, 2, 3])
    with pytest.raises(G.AtomError):
        s.group_create_from_indices("Keep", [n])
    assert s.group_get_n_atoms("Keep") == 3
    # the reference's own clamping of an out-of-range TAIL still works (container.rs:69-73)
    s.group_create_from_indices("Tail", [997, 998, 5000])
    assert s.group_container("Tail").blocks == [(997, 999)]
    np.testing.assert_allclose(s.group_get_center_naive("Tail"), s.get_positions()[997:1000].mean(0), atol=1e-6)
    s.close()


def _blob_pair(G, n, nslots, box):
    rng = np.random.default_rng(5)
    m = rng.uniform(1, 16, n).astype(np.float32)
    base = (rng.normal(0, 0.4, (n, 3)) + np.asarray(box[:3]) / 2).astype(np.float32)
    ref = G.System(n, masses=m, box=box, positions=base)
    cur = G.System(n, masses=m, n_slots=nslots)
    return ref, cur, m, base, rng


def test_double_buffered_uploads_keep_each_frames_own_box(G):
    """NPT-style double buffer: every frame has its own box.  Slot s is re-uploaded as soon as its batch has been collected,
    with NO synchronisation by the caller, while the other slot's batch is in flight; an upload into the in-flight slot
    itself is refused (gr_rmsd_batch_end may still need that frame).  Every frame must be analysed with ITS box: the pinned
    box record of a slot is the source of an asynchronous copy and may only be rewritten once that copy has run."""
    n = 200_000
    ref_box = np.array([6.0, 6.0, 6.0, 0, 0, 0, 0, 0, 0], np.float32)
    ref, cur, m, base, rng = _blob_pair(G, n, 2, ref_box)
    plan = G.RMSDPlan(ref, cur, "all")
    idx = np.arange(n)
    nfr = 12
    boxes = [np.array([6.0 + 0.37 * (k % 5), 6.0 - 0.21 * (k % 3), 5.5 + 0.4 * (k % 4), 0, 0, 0, 0, 0, 0], np.float32) for k in range(nfr)]
    hosts = [G.pinned_array((n, 3)) for _ in range(2)]
    frames, want = [], []
    with O.acc64():
        for k in range(nfr):
            fr = O.translate(base + rng.normal(0, 0.03, base.shape).astype(np.float32), idx, rng.uniform(-9, 9, 3), boxes[k])
            frames.append(fr)
            want.append(O.calc_rmsd(base, m, idx, ref_box, fr, m, idx, boxes[k])[0])
    got = []
    hosts[0][0][:] = frames[0]
    cur.upload_async(hosts[0][0], boxes[0], 0)
    for k in range(nfr):
        s = k % 2
        plan.begin(s, 1, fit=False)
        with pytest.raises(G.DeviceError) as e:
            cur.upload_async(hosts[s][0], boxes[k], s)                    # the in-flight slot: refused, nothing changes
        assert e.value.status == G._lib.E_INVALID_ARG
        if k + 1 < nfr:
            cur.upload_wait(1 - s)                                        # the staging buffer of the other slot is free again
            hosts[1 - s][0][:] = frames[k + 1]
            cur.upload_async(hosts[1 - s][0], boxes[k + 1], 1 - s)        # beside the kernels of frame k
        r, st = plan.end()
        assert st[0] == 0
        got.append(float(r[0]))
    assert np.abs(np.array(got) - np.array(want)).max() <= 1e-5, (got, want)
    plan.close(); ref.close(); cur.close()
    for _, p in hosts:
        G.pinned_free(p)


def test_context_refuses_everything_but_uploads_while_a_batch_is_in_flight(G):
    n = 5000
    box = np.array([6.0, 6.0, 6.0, 0, 0, 0, 0, 0, 0], np.float32)
    ref, cur, m, base, rng = _blob_pair(G, n, 2, box)
    cur.set_frame(base, box, slot=0); cur.set_frame(base, box, slot=1)
    cur.group_create_from_ranges("G", [(0, 999)]); ref.group_create_from_ranges("G", [(0, 999)])
    plan, plan2 = G.RMSDPlan(ref, cur, "G"), G.RMSDPlan(ref, cur, "all")
    want, _ = plan.rmsd(0, 1)
    plan.begin(0, 1, fit=False)
    for call in (lambda: cur.group_create_from_ranges("G", [(0, 1999)]), lambda: cur.group_remove("G"), lambda: cur.set_masses(m * 2),
                 lambda: cur.group_get_com("G"), lambda: plan2.rmsd(1, 1), lambda: plan2.begin(1, 1, fit=False), lambda: cur.atoms_wrap()):
        with pytest.raises(G.GroanError) as e:
            call()
        assert e.value.status == G._lib.E_INVALID_ARG
    cur.upload_async(base, box, 1); cur.upload_wait(1)        # uploads are what the window is for
    r, st = plan.end()
    assert st[0] == 0 and r[0] == want[0]
    # afterwards everything works again, and a changed group / changed masses are picked up by the plan (weights stay the
    # REFERENCE's masses, rmsd.rs:154-155: the "weights == target masses" shortcut must be re-decided)
    cur.set_masses(m * 2)
    r2, st = plan.rmsd(0, 1)
    assert st[0] == 0 and abs(float(r2[0]) - float(want[0])) <= 1e-6
    cur.group_create_from_ranges("G", [(0, 1999)])
    with pytest.raises(G.RMSDError) as e:
        plan.rmsd(0, 1)
    assert e.value.variant == "InconsistentGroup" and e.value.detail[1:] == (1000, 2000)
    plan.close(); plan2.close(); ref.close(); cur.close()
