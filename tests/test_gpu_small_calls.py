"""One-frame calls on small selections (gr_small.h, GR_TUNE_SMALL_CALLS): ONE single-wave dispatch whose result the host reads out of
coherent host memory -- the reference's per-frame `FrameAnalyze::analyze` on a protein (traj_convert.rs:76-83; System::group_get_com,
group_estimate_com, group_get_center, calc_rmsd: analysis.rs:52-320, rmsd.rs:75-129).  Checked here: the single-wave kernels against
the oracle and against the batched kernels (GR_TUNE_SMALL_CALLS = 0) on the same frames, contiguous and scattered selections, three
kinds of cell, the reference's error order for atoms without position / mass, a frame whose image proof fails (the usual redo), a
batch against its per-frame calls bit for bit, and that the path is the one that ran (GR_STAT_SMALL_CALLS)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


@pytest.fixture(scope="module", autouse=True)
def warm_single_wave_kernels(G):
    """Every single-wave kernel variant once, untimed, on a throwaway context: the FIRST launch of a kernel in a process (code-object load
    on a cold box) can outlast the 20 ms the host polls for a small call's result -- the call then synchronises the stream instead and
    counts it (GR_STAT_SMALL_SYNC_FALLBACKS).  With the kernels loaded, the tests below assert that counter stays 0 on their own
    contexts: the polling path must be the one the tests exercise (round 4 had dropped the assertion instead)."""
    box = O.box_from_lengths_angles([6.0, 6.0, 6.0], [60.0, 60.0, 90.0])
    s, frames, m, rng = make(G, 2000, 2, box, 1)
    s.group_create_from_ranges("a", [(10, 300)]); s.group_create_from_indices("b", np.unique(rng.integers(0, 2000, 200)))
    ref = G.System(2000, masses=m, box=box, positions=frames[2])
    ref.group_create_from_ranges("a", [(10, 300)]); ref.group_create_from_indices("b", np.unique(rng.integers(0, 2000, 200)))
    for name in ("a", "b"):
        for fn in ("group_get_com", "group_get_center", "group_estimate_com", "group_estimate_center", "group_get_com_naive", "group_get_center_naive"):
            if hasattr(s, fn):
                getattr(s, fn)(name, slot=0)
        s.group_get_com_batch(name, 0, 2)
    s.group_distance("a", "b", G.Dimension.XYZ, slot=0)
    plan = G.RMSDPlan(ref, s, "a")
    plan.rmsd(0, 1); plan.rmsd(0, 2); plan.rmsd_fit(0, 1); plan.rmsd_fit(0, 2)
    plan.close(); ref.close(); s.close()
    yield


BOXES = {
    "orthorhombic": ([6.44, 6.76, 7.26], [90.0, 90.0, 90.0]),
    "dodecahedron": ([7.0, 7.0, 7.0], [60.0, 60.0, 90.0]),
    "triclinic": ([7.5, 7.0, 6.5], [75.0, 80.0, 70.0]),
}


def make(G, n, nf, box, seed, spread=0.5):
    rng = np.random.default_rng(seed)
    masses = rng.uniform(1.0, 16.0, n).astype(np.float32)
    s = G.System(n, masses=masses, n_slots=nf + 1)
    frames = []
    for f in range(nf + 1):
        blob = rng.normal(0, spread, (n, 3)) + rng.uniform(0.1, 0.9, 3) * np.array([box[0], box[1], box[2]])
        pos = O.wrap_atoms(blob.astype(np.float32), np.arange(n), box)
        s.set_frame(pos, box, slot=f); frames.append(pos)
    return s, frames, masses, rng


@pytest.mark.parametrize("cell", list(BOXES))
def test_centres_of_small_selections(G, cell):
    box = O.box_from_lengths_angles(*BOXES[cell])
    n, nf = 5000, 4
    s, frames, m, rng = make(G, n, nf, box, 11)
    sels = {"block": np.arange(37, 400), "scattered": np.unique(rng.integers(0, n, 300)), "one": np.array([123]), "edge": np.arange(0, 4096)}
    s.group_create_from_ranges("block", [(37, 399)])
    s.group_create_from_indices("scattered", sels["scattered"])
    s.group_create_from_ranges("one", [(123, 123)])
    s.group_create_from_ranges("edge", [(0, 4095)])
    calls = (("group_get_com", O.get_center, True), ("group_get_center", O.get_center, False), ("group_estimate_com", O.estimate_center, True),
             ("group_estimate_center", O.estimate_center, False), ("group_get_com_naive", lambda pos, idx, box, mass=None: O.center_naive(pos, idx, mass=mass), True))
    for name, idx in sels.items():
        for f in range(nf):
            for fn, orc, w in calls:
                before = s.stat("small_calls")
                got = getattr(s, fn)(name, slot=f)
                ran_small = s.stat("small_calls") == before + 1
                # (get_com / get_center of 4096 atoms is the one-pass centre's, every other call here the single wave's)
                assert ran_small == (not (name == "edge" and fn in ("group_get_com", "group_get_center"))), (name, fn)
                with O.acc64():
                    want = orc(frames[f], idx, box, mass=m if w else None)
                assert np.abs(got - want).max() <= 1e-5, (name, fn, f, got, want)
                s.set_tuning(small_calls=0)
                ref = getattr(s, fn)(name, slot=f)
                s.set_tuning(small_calls=4096)
                assert np.abs(got - ref).max() <= 2e-6, (name, fn, f, got, ref)
    assert s.stat("small_sync_fallbacks") == 0          # (the kernels were loaded by warm_single_wave_kernels: every call above was polled out)
    # a batch of such frames equals its per-frame calls bit for bit (one wave per frame, the same kernel stages)
    for name in ("block", "scattered"):
        got, st = s.group_get_com_batch(name, 0, nf)
        assert (st == 0).all()
        for f in range(nf):
            assert np.array_equal(got[f], s.group_get_com(name, slot=f))
        got, st = s.group_estimate_com_batch(name, 0, nf)
        for f in range(nf):
            assert np.array_equal(got[f], s.group_estimate_com(name, slot=f))
    s.close()


@pytest.mark.parametrize("cell", list(BOXES))
def test_rmsd_of_small_selections(G, cell):
    box = O.box_from_lengths_angles(*BOXES[cell])
    n, nf = 5000, 5
    s, frames, m, rng = make(G, n, nf, box, 23, spread=0.35)
    ref = G.System(n, masses=m, box=box, positions=frames[nf])
    scattered = np.unique(rng.integers(0, n, 500))
    for x in (ref, s):
        x.group_create_from_ranges("block", [(100, 462)])
        x.group_create_from_indices("scattered", scattered)
    for name, idx in (("block", np.arange(100, 463)), ("scattered", scattered)):
        plan = G.RMSDPlan(ref, s, name)
        singles = []
        for f in range(nf):
            before = s.stat("small_calls")
            r, st = plan.rmsd(f, 1)
            assert s.stat("small_calls") == before + 1
            assert st[0] == 0
            with O.acc64():
                want = O.calc_rmsd(frames[nf], m, idx, box, frames[f], m, idx, box)[0]
            assert abs(float(r[0]) - want) <= 1e-5, (name, f, float(r[0]), want)
            s.set_tuning(small_calls=0)
            r0, _ = plan.rmsd(f, 1)
            s.set_tuning(small_calls=4096)
            assert abs(float(r[0]) - float(r0[0])) <= 2e-6
            singles.append(r[0])
        rb, st = plan.rmsd(0, nf)
        assert (st == 0).all() and np.array_equal(rb, np.array(singles, np.float32))
        # ... and with the fit (calc_rmsd_and_fit, rmsd.rs:141-166, 508-528): the same rmsd, every atom of the frame transformed; one
        # frame per call on a twin system == the batch, bit for bit
        twin = G.System(n, masses=m, n_slots=nf + 1)
        for f in range(nf):
            twin.set_frame(frames[f], box, slot=f)
        if name == "block": twin.group_create_from_ranges(name, [(100, 462)])
        else: twin.group_create_from_indices(name, scattered)
        plan2 = G.RMSDPlan(ref, twin, name)
        before = twin.stat("small_calls")
        for f in range(nf):
            r1, st1 = plan2.rmsd_fit(f, 1)
            assert st1[0] == 0 and r1[0] == singles[f]
        assert twin.stat("small_calls") == before + nf
        rb2, st2 = plan.rmsd_fit(0, nf)
        assert (st2 == 0).all() and np.array_equal(rb2, rb)
        for f in range(nf):
            assert np.array_equal(twin.get_positions(f), s.get_positions(f))
        with O.acc64():
            ro, want = O.calc_rmsd_and_fit(frames[nf], m, idx, box, frames[0], m, idx, box)
        assert abs(float(rb2[0]) - ro) <= 1e-5 and np.abs(s.get_positions(0) - want).max() <= 5e-5
        for f in range(nf):
            s.set_frame(frames[f], box, slot=f)          # (the next selection starts from the unfitted frames again)
        plan2.close(); twin.close()
        plan.close()
    assert s.stat("small_sync_fallbacks") == 0
    ref.close(); s.close()


def test_errors_in_the_reference_order_and_a_frame_whose_proof_fails(G):
    box = O.box_from_lengths_angles(*BOXES["orthorhombic"])
    n = 3000
    s, frames, m, rng = make(G, n, 2, box, 5)
    s.group_create_from_ranges("block", [(10, 500)])
    # an atom without position, another without mass: estimate_com tests the mass first (iterators.rs:1324-1339), get_com every position
    # before any mass (:1405-1422), the naive centre atom by atom, the position before the mass (:946-958): atom 100 has a position, its mass is the first thing missing
    pos = frames[0].copy(); pos[200] = np.nan
    s.set_frame(pos, box, slot=0)
    mm = m.copy(); mm[100] = np.nan
    s.set_masses(mm)
    for fn, variant, idx in (("group_estimate_com", "InvalidMass", 100), ("group_get_com", "InvalidPosition", 200), ("group_get_com_naive", "InvalidMass", 100),
                             ("group_get_center", "InvalidPosition", 200)):
        for small in (4096, 0):
            s.set_tuning(small_calls=small)
            with pytest.raises(G.GroupError) as e:
                getattr(s, fn)("block", slot=0)
            assert e.value.variant == variant and e.value.detail == idx, (fn, small, e.value.variant, e.value.detail)
    s.set_tuning(small_calls=4096)
    s.set_masses(m)
    s.set_frame(frames[0], box, slot=0)
    # RMSD of a selection that spans more than half the cell: the single pass's image proof fails, the frame is redone on the literal path
    wide = frames[1].copy()
    wide[10:501, 0] = np.linspace(0.2, box[0] * 0.75, 491).astype(np.float32)
    s.set_frame(wide, box, slot=1)
    ref = G.System(n, masses=m, box=box, positions=frames[0])
    ref.group_create_from_ranges("block", [(10, 500)])
    plan = G.RMSDPlan(ref, s, "block")
    r, st = plan.rmsd(1, 1)
    idx = np.arange(10, 501)
    with O.acc64():
        want = O.calc_rmsd(frames[0], m, idx, box, wide, m, idx, box)[0]
    assert st[0] == 0 and abs(float(r[0]) - want) <= 1e-5
    assert plan.last_fallbacks() == 1
    plan.close(); ref.close(); s.close()


@pytest.mark.parametrize("cell", list(BOXES))
def test_group_distance_of_two_small_groups_is_one_dispatch(G, cell):
    """group_distance (analysis.rs:348-360): the centres of both groups in one launch of two waves; the same numbers as the two centre calls,
    the oracle's distance, errors of the first group before the second's"""
    box = O.box_from_lengths_angles(*BOXES[cell])
    n = 6000
    s, frames, m, rng = make(G, n, 2, box, 31, spread=0.4)
    ia, ib = np.arange(50, 413), np.unique(rng.integers(2000, 5000, 200))
    s.group_create_from_ranges("a", [(50, 412)])
    s.group_create_from_indices("b", ib)
    for f in range(2):
        ca, cb = s.group_get_center("a", slot=f), s.group_get_center("b", slot=f)
        for dim in (G.Dimension.XYZ, G.Dimension.X, G.Dimension.YZ):
            before = s.stat("small_calls")
            d = s.group_distance("a", "b", dim, slot=f)
            assert s.stat("small_calls") == before + 1
            s.set_tuning(small_calls=0)
            d0 = s.group_distance("a", "b", dim, slot=f)
            s.set_tuning(small_calls=4096)
            assert abs(d - d0) <= 2e-6
            with O.acc64():
                want = O.distance(O.get_center(frames[f], ia, box), O.get_center(frames[f], ib, box), dim.name.lower(), box)
            assert abs(d - want) <= 2e-5, (cell, f, dim, d, want)
            assert abs(d - O.distance(ca, cb, dim.name.lower(), box)) <= 2e-6
    pos = frames[0].copy(); pos[60] = np.nan; pos[int(ib[3])] = np.nan
    s.set_frame(pos, box, slot=0)
    with pytest.raises(G.GroupError) as e:
        s.group_distance("a", "b", G.Dimension.XYZ, slot=0)
    assert e.value.variant == "InvalidPosition" and e.value.detail == 60
    with pytest.raises(G.GroupError) as e:
        s.group_distance("b", "a", G.Dimension.XYZ, slot=0)
    assert e.value.variant == "InvalidPosition" and e.value.detail == int(ib[3])
    s.close()
