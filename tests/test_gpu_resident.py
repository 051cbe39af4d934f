"""The resident RMSD-fit pass (gr_resident.h: one launch, every frame read once and written once, the frame
waiting on chip for its rotation) against the oracle and against the two-pass path on the same frames.  The pass is the
default for frames that fill the chip (tests/test_gpu_resident_fullsize.py runs it at the benchmark's shape);
GR_TUNE_RESIDENT = 2 forces it here for systems far smaller than the chip: few streaming workgroups, a ragged last one, idle
waves, both kernels (whole-system selections park image vectors, any other selection parks rows), short batches, failed
frames, a box per frame, a launch that never starts and a launch that is aborted from inside."""
import numpy as np
import pytest

import oracle_lib as O
from groan_rs_amd import workload as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def _systems(G, n, nf, box, sel):
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, 0.2 * min(box[0], box[1], box[2]), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.04, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    for s in (ref, cur):
        s.group_create_from_ranges("S", [sel])
    frames = [cur.get_positions(f) for f in range(nf)]
    return masses, cur, ref, ref_pos, frames


@pytest.mark.parametrize("n,sel", [(70_001, (0, 70_000)), (70_001, (1003, 69_990)), (70_001, (0, 69_990)), (5_000, (17, 4_000)), (5_000, (0, 4_999)), (300, (0, 299))])
@pytest.mark.parametrize("tric,streams", [(False, 1), (True, 1), (True, 3)])
def test_resident_matches_oracle_and_two_pass(G, n, sel, tric, streams):
    nf = 11                                             # odd, longer than the pipeline (6 frames between sums and fit); 3 streams: 4 + 4 + 3 turns
    box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]) if tric else W.box_from_lengths_angles([7.0, 6.5, 6.0], [90.0, 90.0, 90.0])
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, sel)
    idx = np.arange(sel[0], sel[1] + 1)
    with O.acc64():
        want = [O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box) for f in range(nf)]
    plan = G.RMSDPlan(ref, cur, "S")
    got = {}
    for mode in (0, 2):
        cur.set_tuning(resident=mode, resident_streams=streams)
        cur.profile_enable(True)
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        r, st = plan.rmsd_fit(0, nf)
        assert (st == 0).all() and plan.last_fallbacks() == 0, mode
        prof = cur.profile_read()
        assert (prof["k_fit_resident"][1] > 0) == (mode == 2), (mode, prof)      # the pass that was asked for is the one that ran
        assert mode == 0 or cur.stat("res_last_streams") == streams
        got[mode] = (np.array(r), [cur.get_positions(f) for f in range(nf)])
        for f in range(nf):
            assert abs(float(r[f]) - want[f][0]) <= 1e-5, (mode, f, float(r[f]), want[f][0])
            assert np.abs(got[mode][1][f] - want[f][1]).max() <= 5e-5, (mode, f)
    assert np.abs(got[0][0] - got[2][0]).max() <= 2e-6
    for f in range(nf):
        assert np.abs(got[0][1][f] - got[2][1][f]).max() <= 2e-5
    # the same launch again gives the same bits (order of arrival inside the launch does not reach the results)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r2, _ = plan.rmsd_fit(0, nf)
    assert np.array_equal(np.array(r2), got[2][0])
    for f in range(nf):
        assert np.array_equal(cur.get_positions(f), got[2][1][f])
    plan.close(); ref.close(); cur.close()


def test_resident_short_batches_and_failed_frames(G):
    """1, 2, 3 and 4 frames (shorter than, equal to and just longer than the pipeline); a frame without a box and a frame with a
    missing position fail exactly as on the two-pass path and are left unmodified, the frames around them are fitted."""
    n, nf = 20_000, 8
    box = W.box_from_lengths_angles([6.0, 6.0, 6.0], [90.0, 90.0, 90.0])
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, (0, n - 1))
    plan = G.RMSDPlan(ref, cur, "S")
    res = {}
    for mode in (0, 2):
        cur.set_tuning(resident=mode)
        for nb in (1, 2, 3, 4, 7):
            for f in range(nb):
                cur.set_frame(frames[f], box, slot=f)
            r, st = plan.rmsd_fit(0, nb)
            assert (st == 0).all()
            res[(mode, nb)] = (np.array(r), [cur.get_positions(f) for f in range(nb)])
        bad = frames[2].copy(); bad[777] = np.nan
        for f in range(nf):
            cur.set_frame(bad if f == 2 else frames[f], box, slot=f)
        r, st = plan.rmsd_fit(0, nf, raise_on_error=False)
        assert st[2] != 0 and all(st[f] == 0 for f in range(nf) if f != 2), (mode, st)
        after = [cur.get_positions(f) for f in range(nf)]
        assert np.array_equal(np.nan_to_num(after[2], nan=-1.0), np.nan_to_num(bad, nan=-1.0)), mode        # the failed frame is untouched
        res[(mode, "bad")] = (np.array(r), np.array(st), after)
    for nb in (1, 2, 3, 4, 7):
        assert np.abs(res[(0, nb)][0] - res[(2, nb)][0]).max() <= 2e-6
        for f in range(nb):
            assert np.abs(res[(0, nb)][1][f] - res[(2, nb)][1][f]).max() <= 2e-5
    assert np.array_equal(res[(0, "bad")][1], res[(2, "bad")][1])
    for f in range(nf):
        if f != 2:
            assert abs(res[(0, "bad")][0][f] - res[(2, "bad")][0][f]) <= 2e-6
            assert np.abs(res[(0, "bad")][2][f] - res[(2, "bad")][2][f]).max() <= 2e-5
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("whole", [False, True])
def test_frame_streams_do_not_reach_the_results(G, whole):
    """frames that fill a fraction of the chip run as several frame streams side by side in one launch (stream s of S: frames s,
    s + S, ...; the finalizer workgroups then close 2, 4 or 8 frames at a time, one per team of waves).  Which stream a frame
    rides in must not show in its results: 1, 2, 3, 4, 8 and 13 streams give the same bits, for segments that give every stream
    the same number of turns and for ragged ones (13 = 4 + 3 + 3 + 3 frames; 3 frames with room for 4 streams), with a failed
    frame in the middle and a box per frame"""
    n, nf = 20_000, 13
    masses = W.masses_cycle(n)
    boxes = [W.box_from_lengths_angles([6.0 + 0.01 * f, 6.0, 6.0 - 0.004 * f], [80.0, 85.0 + 0.1 * f, 75.0]) for f in range(nf)]
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, boxes[0], 1.1, W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.04, W.SEED)
    ref = G.System(n, masses=masses, box=boxes[0], positions=cur.get_positions(nf))
    for s_ in (ref, cur):
        s_.group_create_from_ranges("S", [(0, n - 1) if whole else (3, n - 2)])
    frames = [cur.get_positions(f) for f in range(nf)]
    frames[6] = frames[6].copy(); frames[6][4321] = np.nan
    plan = G.RMSDPlan(ref, cur, "S")
    res = {}
    for streams in (1, 2, 3, 4, 8, 13):
        cur.set_tuning(resident=2, resident_streams=streams)
        for nb in (nf, 3):
            for f in range(nb):
                cur.set_frame(frames[f], boxes[f], slot=f)
            r, st = plan.rmsd_fit(0, nb, raise_on_error=False)
            assert cur.stat("res_last_streams") == min(streams, nb)
            assert [f for f in range(nb) if st[f] != 0] == ([6] if nb > 6 else []), (streams, nb, st)
            res[(streams, nb)] = (np.array(r), [cur.get_positions(f) for f in range(nb)])
    for streams in (2, 3, 4, 8, 13):
        for nb in (nf, 3):
            assert np.array_equal(res[(streams, nb)][0], res[(1, nb)][0], equal_nan=True), (streams, nb)
            for f in range(nb):
                assert np.array_equal(res[(streams, nb)][1][f], res[(1, nb)][1][f], equal_nan=True), (streams, nb, f)
    assert cur.stat("res_launches") == 12 and cur.stat("res_aborts") == 0
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("whole", [False, True])
def test_workgroups_of_fewer_than_1024_groups_give_the_same_results(G, whole):
    """a frame that would leave CUs idle when cut into 1024-group workgroups is cut into smaller ones (GR_TUNE_RESIDENT_WG_GROUPS: 576,
    768, 960 groups -- the lanes' second groups only in the first waves -- and 320: no second groups at all, three idle waves):
    the per-workgroup partial sums change, the results stay within the tolerance against the two-pass path and the oracle"""
    n, nf = 41_111, 9
    box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [75.0, 80.0, 70.0])
    sel = (0, n - 1) if whole else (700, n - 4)
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, sel)
    idx = np.arange(sel[0], sel[1] + 1)
    with O.acc64():
        want = [O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box) for f in range(nf)]
    plan = G.RMSDPlan(ref, cur, "S")
    for gwg, streams in ((1024, 1), (960, 1), (768, 2), (576, 1), (320, 3)):
        cur.set_tuning(resident=2, resident_streams=streams, resident_wg_groups=gwg)
        cur.profile_enable(True)
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        r, st = plan.rmsd_fit(0, nf)
        assert (st == 0).all() and cur.profile_read()["k_fit_resident"][1] == 1, gwg
        for f in range(nf):
            assert abs(float(r[f]) - want[f][0]) <= 1e-5, (gwg, f, float(r[f]), want[f][0])
            assert np.abs(cur.get_positions(f) - want[f][1]).max() <= 5e-5, (gwg, f)
    with pytest.raises(Exception):
        cur.set_tuning(resident_wg_groups=100)          # not a multiple of 64
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("whole", [False, True])
def test_resident_with_a_different_box_in_every_frame(G, whole):
    """constant-pressure runs: every frame has its own box -> the kernel variant that reads the box per frame"""
    n, nf = 30_000, 9
    masses = W.masses_cycle(n)
    boxes = [W.box_from_lengths_angles([7.0 + 0.01 * f, 6.5 - 0.005 * f, 6.0 + 0.002 * f], [75.0, 80.0 + 0.1 * f, 70.0]) for f in range(nf)]
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, boxes[0], 1.2, W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.04, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=boxes[0], positions=ref_pos)
    for s_ in (ref, cur):
        s_.group_create_from_ranges("S", [(0, n - 1) if whole else (5, n - 9)])
    frames = [cur.get_positions(f) for f in range(nf)]
    idx = np.arange(n) if whole else np.arange(5, n - 8)
    with O.acc64():
        want = [O.calc_rmsd_and_fit(ref_pos, masses, idx, boxes[0], frames[f], masses, idx, boxes[f]) for f in range(nf)]
    plan = G.RMSDPlan(ref, cur, "S")
    cur.set_tuning(resident=2)
    cur.profile_enable(True)
    for f in range(nf):
        cur.set_frame(frames[f], boxes[f], slot=f)
    r, st = plan.rmsd_fit(0, nf)
    assert (st == 0).all() and cur.profile_read()["k_fit_resident"][1] == 1
    for f in range(nf):
        assert abs(float(r[f]) - want[f][0]) <= 1e-5, (f, float(r[f]), want[f][0])
        assert np.abs(cur.get_positions(f) - want[f][1]).max() <= 5e-5, f
    plan.close(); ref.close(); cur.close()


def test_resident_launch_that_never_starts_falls_back_cleanly(G):
    """a launch whose workgroups do not all get onto the chip (a shared device) must leave every frame untouched; the segment then
    runs on the two-pass path and the context stops choosing the resident pass"""
    n, nf = 20_000, 24
    box = W.box_from_lengths_angles([6.0, 6.0, 6.0], [90.0, 90.0, 90.0])
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, (0, n - 1))
    plan = G.RMSDPlan(ref, cur, "S")
    cur.set_tuning(resident=0)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    want_r, st = plan.rmsd_fit(0, nf)
    want = [cur.get_positions(f) for f in range(nf)]
    cur.set_tuning(resident=2, test_resident_no_start=1)
    cur.profile_enable(True)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r, st = plan.rmsd_fit(0, nf)
    prof = cur.profile_read()
    assert (st == 0).all() and prof["k_fit_pk"][1] > 0                     # the two-pass kernels did the work
    assert np.array_equal(np.array(r), np.array(want_r))
    for f in range(nf):
        assert np.array_equal(cur.get_positions(f), want[f])
    assert cur.stat("res_handshake_misses") == 1 and cur.stat("res_max_wgs") > 0
    # the context sits out the next segments (4 after the first miss), then tries the pass again -- and keeps it when it starts
    ran = []
    for _ in range(6):
        cur.profile_enable(True)
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        r, st = plan.rmsd_fit(0, nf)
        assert (st == 0).all()
        ran.append(cur.profile_read()["k_fit_resident"][1])
    assert ran == [0, 0, 0, 0, 1, 1], ran
    assert cur.stat("res_launches") == 2
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("whole,streams", [(True, 1), (False, 1), (True, 3)])
@pytest.mark.parametrize("at", [0, 3, 9, 22])
def test_resident_launch_aborted_from_inside_loses_nothing(G, whole, streams, at):
    """a wait that runs out of patience aborts the launch from inside (here: the finalizer of frame `at` raises the abort
    instead of closing its frame).  Frames the launch had completed keep their results, every other frame is still untouched
    and is redone on the two-pass path: the call succeeds, every frame is fitted exactly once, nothing is torn"""
    n, nf = 20_000, 24
    box = W.box_from_lengths_angles([6.0, 6.0, 6.0], [60.0, 60.0, 90.0])
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, (0, n - 1) if whole else (40, n - 3))
    plan = G.RMSDPlan(ref, cur, "S")
    cur.set_tuning(resident=0)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    want_r, st = plan.rmsd_fit(0, nf)
    want = [cur.get_positions(f) for f in range(nf)]
    cur.set_tuning(resident=2, resident_streams=streams, test_resident_abort_at=at)
    cur.profile_enable(True)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r, st = plan.rmsd_fit(0, nf)
    prof = cur.profile_read()
    assert (st == 0).all(), st
    assert prof["k_fit_resident"][1] == 1 and cur.stat("res_aborts") == 1 and cur.stat("res_launches") == 0
    redone = cur.stat("res_redone_frames")
    # frame `at` itself and whatever had not been fitted when the grid drained: the frames behind it and, with several streams,
    # frames of the other streams a little ahead of it
    assert 1 <= redone <= (nf - at if streams == 1 else nf), (at, redone)
    assert (prof["k_fit_pk"][2] == redone), (prof, redone)
    assert np.abs(np.array(r) - np.array(want_r)).max() <= 2e-6
    for f in range(nf):
        assert np.abs(cur.get_positions(f) - want[f]).max() <= 2e-5, f
    # the next launch runs to the end
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r, st = plan.rmsd_fit(0, nf)
    assert (st == 0).all() and cur.stat("res_launches") == 1
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("whole", [True, False])
@pytest.mark.parametrize("cell,streams", [("ortho", 1), ("tric", 1), ("tric", 3), ("dodeca", 8)])
def test_frames_whose_image_proof_fails_inside_a_resident_launch(G, whole, cell, streams):
    """The switch-off of the shortcut the resident pass rests on.  The V kernel parks image vectors and fits with R v + t0, which is
    the path's q only "once the frame's image proof holds" (gr_resident.h); a frame whose proof FAILS -- a group wider than half
    the cell: stretched, or two lobes -- must come out of the launch untouched (status GR_ST_FALLBACK published by its finalizer,
    the fit stage skipped) and be redone by the literal multi-pass path (rmsd.rs:425-446,508-528), without disturbing the frames
    that share its stream, its finalizer round and its LDS / register slots.  Frames 0, 6 (the first frame fitted after the
    pipeline has filled) and the last one are such frames; both kernel variants; 1, 3 and 8 streams (8: the finalizer teams close
    several frames per round, proof-failing and ordinary ones side by side).  Every frame against the oracle."""
    n, nf = 20_000, 26
    box = {"ortho": W.box_from_lengths_angles([6.0, 6.0, 6.0], [90.0, 90.0, 90.0]), "tric": W.box_from_lengths_angles([7.0, 6.5, 6.0], [75.0, 80.0, 70.0]),
           "dodeca": W.c4_box(6.5)}[cell]
    sel = (0, n - 1) if whole else (40, n - 3)
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, sel)
    bad = {0: "stretched", 6: "two_lobes", nf - 1: "stretched"}
    for f, kind in bad.items():
        frames[f] = W.proof_failing_frame(ref_pos, box, kind, 100 + f)
    idx = np.arange(sel[0], sel[1] + 1)
    with O.acc64():
        want = [O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box) for f in range(nf)]
    plan = G.RMSDPlan(ref, cur, "S")
    cur.set_tuning(resident=2, resident_streams=streams)
    cur.profile_enable(True)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r, st = plan.rmsd_fit(0, nf)
    prof = cur.profile_read()
    assert (st == 0).all(), st
    assert prof["k_fit_resident"][1] == 1 and prof["k_fit_resident"][2] == nf, prof          # ONE resident launch saw all 26 frames
    assert plan.last_fallbacks() == len(bad), plan.last_fallbacks()                           # ... and handed back exactly the three
    assert cur.stat("res_aborts") == 0 and cur.stat("res_launches") == 1 and cur.stat("res_last_streams") == streams
    for f in range(nf):
        assert abs(float(r[f]) - want[f][0]) <= 1e-5, (f, f in bad, float(r[f]), want[f][0])
        assert np.abs(cur.get_positions(f) - want[f][1]).max() <= 5e-5, (f, f in bad)
    # the RMSD alone (no fit) over the same frames: the frames are left as they are, the same three are handed back
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r2, st2 = plan.rmsd(0, nf)
    assert (st2 == 0).all() and plan.last_fallbacks() == len(bad)
    for f in range(nf):
        assert abs(float(r2[f]) - want[f][0]) <= 1e-5, (f, float(r2[f]), want[f][0])
        assert np.array_equal(cur.get_positions(f), frames[f]), f
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("whole", [True, False])
@pytest.mark.parametrize("n,streams,wide", [(70_001, 1, (12, 17)), (20_000, 3, (9, 14))])
def test_aborted_launch_whose_redone_run_holds_a_proof_failing_frame(G, whole, n, streams, wide):
    """the frames an aborted launch never touched are redone as a nested two-pass segment; when that run contains a frame
    whose image proof fails, the nested call's own exact-path redo used to overwrite the scratch records the outer call then read
    (the run's first frame silently got another frame's rmsd).  The finalizer of frame 9 raises the abort; its workgroup then
    publishes every frame it owns as ABORTED and the host redoes those:
      70 001 atoms, 1 stream   one frame per finalizer round: frames 9 and 17 (= 9 + 8 finalizers) are redone, 17 is proof-failing;
                               the system's last workgroup has 128 of 1024 groups, i.e. 6 idle waves whose progress words the
                               host must skip (they used to read "all turns done" and turn untouched frames into torn ones)
      20 000 atoms, 3 streams  two frames per round: the run [8, 9] is redone and its SECOND frame is the proof-failing one."""
    nf = 24
    box = W.box_from_lengths_angles([9.0, 9.0, 9.0], [60.0, 60.0, 90.0])
    sel = (0, n - 1) if whole else (40, n - 3)
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, sel)
    for k, f in enumerate(wide):
        frames[f] = W.proof_failing_frame(ref_pos, box, "two_lobes" if k else "stretched", 200 + f)
    idx = np.arange(sel[0], sel[1] + 1)
    with O.acc64():
        want = [O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box) for f in range(nf)]
    plan = G.RMSDPlan(ref, cur, "S")
    cur.set_tuning(resident=2, resident_streams=streams, test_resident_abort_at=9)
    for f in range(nf):
        cur.set_frame(frames[f], box, slot=f)
    r, st = plan.rmsd_fit(0, nf)
    assert (st == 0).all(), st
    assert cur.stat("res_aborts") == 1 and cur.stat("res_redone_frames") >= 2, cur.stat("res_redone_frames")
    assert plan.last_fallbacks() == 2
    for f in range(nf):
        assert abs(float(r[f]) - want[f][0]) <= 1e-5, (f, float(r[f]), want[f][0])
        assert np.abs(cur.get_positions(f) - want[f][1]).max() <= 5e-5, f
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("mode", [0, 2])
def test_a_group_that_spans_the_cell_fails_the_proof_in_every_frame(G, mode):
    """a membrane-like slab -- the whole cell in x and y -- is wider than half the box in EVERY frame: every frame of the call is handed to
    the literal multi-pass path, which now takes them in runs (one set of launches per run of consecutive frames, not six launches
    and a wait per frame: 89 -> 4.8 us per frame at 2e5 atoms, tools/wide_group_bench.py).  Two-pass path and forced resident launch;
    RMSD, rotation-free parity of the fitted coordinates, statuses; a frame without a position in the middle of the run"""
    n, nf = 30_000, 9
    box = W.box_from_lengths_angles([8.0, 8.0, 6.0], [90.0, 90.0, 90.0])
    rng = np.random.default_rng(12)
    base = np.c_[rng.random(n) * 8.0, rng.random(n) * 8.0, 2.0 + rng.random(n) * 2.0]
    masses = W.masses_cycle(n)
    frames = [W.wrap_into_cell(base + rng.normal(0, 0.04, base.shape) + rng.uniform(0, 8, 3) * [1, 1, 0.2], box) for _ in range(nf)]
    frames[4] = frames[4].copy(); frames[4][777] = np.nan
    ref_pos = base.astype(np.float32)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    cur = G.System(n, masses=masses, n_slots=nf)
    plan = G.RMSDPlan(ref, cur, "all")
    cur.set_tuning(resident=mode)
    idx = np.arange(n)
    for fit in (False, True):
        for f in range(nf):
            cur.set_frame(frames[f], box, slot=f)
        r, st = (plan.rmsd_fit if fit else plan.rmsd)(0, nf, raise_on_error=False)
        assert [f for f in range(nf) if st[f] != 0] == [4], st
        assert plan.last_fallbacks() == nf, plan.last_fallbacks()              # (the frame without a position poisons its sums: the literal path names the atom)
        with O.acc64():
            for f in (0, 3, 5, nf - 1):
                ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box)
                assert abs(float(r[f]) - ro) <= 1e-5, (fit, f, float(r[f]), ro)
                got = cur.get_positions(f)
                assert np.abs(got - (want if fit else frames[f])).max() <= 5e-5, (fit, f)
        assert np.array_equal(np.nan_to_num(cur.get_positions(4), nan=-1.0), np.nan_to_num(frames[4], nan=-1.0))
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("resident", [0, 2])
def test_an_atom_without_position_next_to_the_selections_edge(G, resident):
    """The selection starts and ends in the middle of a 4-atom group, and the atoms of those groups OUTSIDE it have no position:
    they weigh nothing, and their term in sum w |R q - p|^2 must be an exact zero, not 0 * NaN (both passes multiplied by the
    weight only until round 4).  The fit leaves such atoms as they are and moves everything else."""
    n, nf, sel = 20_000, 9, (5, 19_993)
    box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [90.0, 90.0, 90.0])
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, sel)
    cur.set_tuning(resident=resident)
    for f in (2, 7):
        frames[f] = frames[f].copy(); frames[f][4] = np.nan; frames[f][19_994] = np.nan
        cur.set_frame(frames[f], box, slot=f)
    plan = G.RMSDPlan(ref, cur, "S")
    cur.profile_enable(True)
    r, st = plan.rmsd_fit(0, nf, raise_on_error=False)
    assert (cur.profile_read()["k_fit_resident"][1] > 0) == (resident == 2)
    assert (np.array(st) == 0).all(), st
    idx = np.arange(sel[0], sel[1] + 1)
    with O.acc64():
        for f in (1, 2, 7):
            clean = np.nan_to_num(frames[f], nan=1.0)
            ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, clean, masses, idx, box)
            got = cur.get_positions(f)
            assert abs(float(r[f]) - ro) <= 1e-5, (f, float(r[f]), ro)
            fin = np.isfinite(frames[f][:, 0])
            assert np.abs(got[fin] - want[fin]).max() <= 5e-5 and np.isnan(got[~fin]).all()
    plan.close(); ref.close(); cur.close()


def test_step_order_and_metronome_change_nothing_but_the_pace(G):
    """Round 5's two knobs of the resident pass -- the order of a turn (GR_TUNE_RESIDENT_FIT_LAST: fit first / sums first) and the
    metronome of the row requests (GR_TUNE_RESIDENT_METRO_NS: off / a period the launch keeps easily / one it cannot keep) -- move work
    in time, never in value: RMSDs and fitted coordinates of every combination are bit-identical, and the launch reports its pace."""
    n, nf = 70_001, 23
    box = W.box_from_lengths_angles([7.0, 6.5, 6.0], [75.0, 80.0, 70.0])
    masses, cur, ref, ref_pos, frames = _systems(G, n, nf, box, (0, n - 1))
    plan = G.RMSDPlan(ref, cur, "S")
    base = None
    for order in (1, 2):
        for metro in (1, 30_000, 200, 0):
            cur.set_tuning(resident=2, resident_fit_last=order, resident_metro_ns=metro)
            for f in range(nf):
                cur.set_frame(frames[f], box, slot=f)
            r, st = plan.rmsd_fit(0, nf)
            assert (st == 0).all() and cur.stat("res_aborts") == 0
            got = (np.array(r).view(np.uint32), np.stack([cur.get_positions(f) for f in range(nf)]).view(np.uint32))
            assert cur.stat("res_metro_period_ns") == (metro if metro >= 100 else 0) and cur.stat("res_last_turn_ns") > 0 and 500 < cur.stat("res_sclk_mhz") < 3000
            if metro == 30_000:      # a period far above what a turn takes: every slot is waited for, the launch takes as long as the clock says
                assert cur.stat("res_last_turn_ns") >= 24_000 and cur.stat("res_late_permille") < 100
            if metro == 200:         # a period no launch can keep: (nearly) every slot is reached late, nothing waits
                assert cur.stat("res_late_permille") > 500
            if base is None:
                base = got
            assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), (order, metro)
    with pytest.raises(Exception):
        cur.set_tuning(resident_metro_ns=50)            # neither a switch (0, 1) nor a period
    with pytest.raises(Exception):
        cur.set_tuning(resident_fit_last=3)
