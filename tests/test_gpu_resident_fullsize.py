"""The resident RMSD-fit pass (k_fit_resident, gr_resident.h) AT THE SHAPE THE BENCHMARK TIMES: 1e6 atoms in the rhombic
dodecahedron of BASELINE configs[3] (d = 24.18 nm, blob of 0.2 x the shortest height, noise 0.05 nm), DEFAULT tuning, >= 24
frames per call -- the launch of ~245 streaming + 8 finalizer workgroups that occupies every CU, which the small forced
launches of test_gpu_resident.py (1-18 workgroups) never reach: the start handshake with the whole grid, 2^20-range indexing,
the frame pipeline filling (frames 0..5), running (>= 6) and draining (the last 6 frames), the record arrays of a long segment.
Every case asserts that the resident kernel is the one that ran, and compares RMSD (<= 1e-5 nm) and fitted coordinates
(<= 5e-5 nm) with the oracle's restatement of rmsd.rs:425-603 (sums in double: DESIGN.md section 2) on the first frame, frames
on either side of the pipeline depth, a steady-state frame and the last frame."""
import numpy as np
import pytest

import oracle_lib as O
from groan_rs_amd import workload as W

pytestmark = pytest.mark.gpu

N = 1_000_000


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def _check(cur, plan, ref_pos, masses, idx, ref_box, boxes, nf, frames_to_check, noise=0.05):
    """run ONE default-tuned gr_rmsd_fit_batch over nf fresh frames; oracle on the chosen frames"""
    before = {f: cur.get_positions(f) for f in frames_to_check}
    cur.profile_enable(True)
    r, st = plan.rmsd_fit(0, nf)
    prof = cur.profile_read()
    assert (st == 0).all() and plan.last_fallbacks() == 0, (st, plan.last_fallbacks())
    assert np.isfinite(r).all()
    with O.acc64():
        for f in frames_to_check:
            ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, ref_box, before[f], masses, idx, boxes[f])
            assert abs(float(r[f]) - ro) <= 1e-5, (f, float(r[f]), ro)
            got = cur.get_positions(f)
            assert np.abs(got - want).max() <= 5e-5, (f, float(np.abs(got - want).max()))
    return prof, r


def _c4(G, n, nf, sel=None, boxes=None):
    box = W.c4_box()
    masses = W.masses_cycle(n)
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, W.blob_radius(box), W.SEED)
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    name = "all"
    if sel is not None:
        name = "S"
        for s in (ref, cur):
            s.group_create_from_ranges("S", [sel])
    if boxes is not None:
        for f in range(nf):
            cur.set_box(boxes[f], slot=f)
    plan = G.RMSDPlan(ref, cur, name)
    return box, masses, cur, ref, ref_pos, plan


def test_headline_shape_default_tuning_all_atoms(G):
    """bench.py's exact workload (S = N), 32 frames, default tuning: k_fit_resident<.., UBOX = true, 2>"""
    nf = 32
    box, masses, cur, ref, ref_pos, plan = _c4(G, N, nf)
    prof, r = _check(cur, plan, ref_pos, masses, np.arange(N), box, [box] * nf, nf, [0, 5, 6, 17, nf - 1])
    assert prof["k_fit_resident"][1] == 1 and prof["k_fit_resident"][2] == nf, prof      # ONE resident launch did all 32 frames
    assert prof["k_fit_pk"][1] == 0 and prof["k_sums_pk"][1] == 0, prof
    # a second call over fresh copies of the same frames reproduces the bits (arrival order does not reach the results)
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    r2, st2 = plan.rmsd_fit(0, nf)
    assert np.array_equal(np.asarray(r), np.asarray(r2))
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("forced", [False, True])
def test_headline_shape_prefix_selection(G, forced):
    """SURVEY 8(d)'s S = 1e5 prefix variant at full size.  By default a selection of less than 45 % of the system takes the two
    passes (the resident pass advances at the pace of the workgroups that hold the selection: 6.3 us per frame against 4.5);
    forced (GR_TUNE_RESIDENT = 2) it runs the row-parking kernel variant: the first 25 workgroups carry the selection (the last
    of them a ragged end inside a lane's 4-atom group), the other 220 skip the sums arithmetic and only stream + fit"""
    nf, s_last = 24, 99_998
    box, masses, cur, ref, ref_pos, plan = _c4(G, N, nf, sel=(0, s_last))
    if forced:
        cur.set_tuning(resident=2)
    prof, _ = _check(cur, plan, ref_pos, masses, np.arange(s_last + 1), box, [box] * nf, nf, [0, 6, 11, nf - 1])
    assert (prof["k_fit_resident"][1] == 1) == forced and (prof["k_fit_pk"][1] > 0) == (not forced), prof
    plan.close(); ref.close(); cur.close()


def test_nearly_whole_selection_takes_the_row_parking_variant(G):
    """a selection of more than 45 % of the system that is not the whole system: resident by default, kernel variant V = false"""
    nf = 24
    box, masses, cur, ref, ref_pos, plan = _c4(G, N, nf, sel=(1_000, N - 3))
    prof, _ = _check(cur, plan, ref_pos, masses, np.arange(1_000, N - 2), box, [box] * nf, nf, [0, 7, nf - 1])
    assert prof["k_fit_resident"][1] == 1 and prof["k_fit_pk"][1] == 0, prof
    plan.close(); ref.close(); cur.close()


def test_a_plan_destroyed_with_a_batch_in_flight_gives_the_device_back(G):
    """gr_rmsd_batch_begin takes the per-device slot of the resident pass; destroying the plan (or the context) before
    gr_rmsd_batch_end must hand it back, or every later context of the process would silently fall to the two passes"""
    nf = 24
    box, masses, cur, ref, ref_pos, plan = _c4(G, N, nf)
    plan.begin(0, nf, fit=True)
    plan.close()                                                                   # mid-batch
    plan2 = G.RMSDPlan(ref, cur, "all")
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    cur.profile_enable(True)
    r, st = plan2.rmsd_fit(0, nf)
    assert (st == 0).all() and cur.profile_read()["k_fit_resident"][1] == 1
    plan2.begin(0, nf, fit=True)                                                   # ... and the context destroyed mid-batch
    cur.close()
    plan2._plan = None                                                             # (its context is gone: the handle must not be used again)
    box, masses, cur3, ref3, ref_pos3, plan3 = _c4(G, N, nf)
    cur3.profile_enable(True)
    r, st = plan3.rmsd_fit(0, nf)
    assert (st == 0).all() and cur3.profile_read()["k_fit_resident"][1] == 1
    plan3.close(); ref3.close(); cur3.close(); ref.close()


def test_headline_shape_with_a_different_box_in_every_frame(G):
    """constant-pressure run at full size: every frame has its own (slightly breathing) dodecahedron -> UBOX = false, the box
    constants are scalar loads per frame in both stages"""
    nf = 24
    d = [24.18 * (1.0 + 2.0e-4 * ((f * 7) % 11 - 5)) for f in range(nf)]
    boxes = [W.c4_box(x) for x in d]
    box, masses, cur, ref, ref_pos, plan = _c4(G, N, nf, boxes=boxes)
    prof, _ = _check(cur, plan, ref_pos, masses, np.arange(N), box, boxes, nf, [0, 7, 13, nf - 1])
    assert prof["k_fit_resident"][1] == 1 and prof["k_fit_pk"][1] == 0, prof
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("n,nf,streams", [(500_000, 40, 2), (330_000, 50, 3), (250_000, 70, 4), (125_000, 150, 8), (62_000, 260, 15), (20_000, 600, 32)])
def test_frames_that_fill_a_fraction_of_the_chip_run_as_streams_by_default(G, n, nf, streams):
    """DEFAULT tuning, systems of a half, a third, a quarter, an eighth, a sixteenth and 1/50 of the chip: 2, 3, 4, 8, 15, 32 frame streams side
    by side in ONE resident launch (stream s: frames s, s + S, ...; ragged: 50 = 17 + 17 + 16 turns, 70 = 18 + 18 + 17 + 17), every
    stream with 16 frames or more; the finalizer workgroups close 2, 2, 4, 8, 8, 8 frames at a time.  Oracle on frames of every stream, at the start, around the pipeline depth of each stream and at the end."""
    box, masses, cur, ref, ref_pos, plan = _c4(G, n, nf)
    check = sorted({0, 1, streams - 1, 6 * streams - 1, 6 * streams, 6 * streams + 1, nf - streams, nf - 1})
    prof, r = _check(cur, plan, ref_pos, masses, np.arange(n), box, [box] * nf, nf, check)
    assert prof["k_fit_resident"][1] == 1 and prof["k_fit_resident"][2] == nf and prof["k_fit_pk"][1] == 0, prof
    assert cur.stat("res_last_streams") == streams
    # a call that can feed one stream less (16 frames each): the pass with S - 1 streams if they still fill 1/16 of the chip (GR_TUNE_RESIDENT_FILL's
    # default since round 5; 10/16 before), the two passes otherwise -- the same results either way
    cur.synth_frames(nf, 0, nf, 0, 0.05, W.SEED)
    cur.profile_enable(True)
    r2, st2 = plan.rmsd_fit(0, 16 * streams - 1)
    prof = cur.profile_read()
    wgs_less = (n + 4095) // 4096
    if streams - 1 == 1 and wgs_less <= 170:          # ONE stream of a frame that fills half of the chip: workgroups of 768 groups (gr_api.hip resident_wgs)
        wgs_less = ((n + 255) // 256 * 64 + 767) // 768
    still = (streams - 1) * wgs_less * 16 >= cur.stat("res_max_wgs") * 1
    assert (st2 == 0).all() and (prof["k_fit_resident"][1] == 1) == still and (prof["k_fit_pk"][1] > 0) == (not still), prof
    assert not still or cur.stat("res_last_streams") == streams - 1
    assert np.abs(np.asarray(r2) - np.asarray(r)[:16 * streams - 1]).max() <= 2e-6
    plan.close(); ref.close(); cur.close()


def test_largest_frame_the_pass_accepts_and_the_first_it_refuses(G):
    """the `streaming workgroups + 2 finalizers <= workgroups the device holds` boundary (gr_api.hip resident_wgs): the
    largest system takes the pass with only two finalizer workgroups; one workgroup more and the two-pass path runs"""
    nf = 24
    box = W.c4_box()
    probe = G.System(1024, n_slots=1)
    max_wgs = probe.stat("res_max_wgs")
    probe.close()
    assert max_wgs >= 64, max_wgs
    n_in = (max_wgs - 2) * 4096 - 77            # ragged last tile, streaming workgroups = max_wgs - 2
    n_out = (max_wgs - 2) * 4096 + 1            # one more workgroup
    for n, resident in ((n_in, True), (n_out, False)):
        box, masses, cur, ref, ref_pos, plan = _c4(G, n, nf)
        prof, _ = _check(cur, plan, ref_pos, masses, np.arange(n), box, [box] * nf, nf, [0, nf - 1])
        assert (prof["k_fit_resident"][1] == 1) == resident and (prof["k_fit_pk"][1] > 0) == (not resident), (n, prof)
        plan.close(); ref.close(); cur.close()


def test_headline_shape_with_frames_whose_image_proof_fails(G):
    """The benchmark's launch (1e6 atoms, rhombic dodecahedron, default tuning, k_fit_resident<.., V = true>) with three frames whose
    image proof FAILS among the 26 -- frame 0, frame 6 (the first one fitted once the pipeline is full) and the last: a group
    stretched to 0.62 of the cell / two lobes 0.42 a apart (workload.proof_failing_frame).  The shortcut `R v + t0` must switch
    itself off for exactly those frames: the launch leaves them untouched, the host redoes them on the literal path
    (rmsd.rs:425-446,508-528), and their neighbours in the stream, in the finalizer rounds and in the LDS / register slots are
    fitted as if nothing had happened.  Oracle on the three frames and on the frames either side of each."""
    nf = 26
    box, masses, cur, ref, ref_pos, plan = _c4(G, N, nf)
    bad = {0: "stretched", 6: "two_lobes", nf - 1: "stretched"}
    for f, kind in bad.items():
        cur.set_frame(W.proof_failing_frame(ref_pos, box, kind, 300 + f), box, slot=f)
    check = [0, 1, 5, 6, 7, nf - 2, nf - 1]
    before = {f: cur.get_positions(f) for f in check}
    cur.profile_enable(True)
    r, st = plan.rmsd_fit(0, nf)
    prof = cur.profile_read()
    assert (st == 0).all(), st
    assert prof["k_fit_resident"][1] == 1 and prof["k_fit_resident"][2] == nf, prof
    assert plan.last_fallbacks() == len(bad) and cur.stat("res_aborts") == 0, (plan.last_fallbacks(), cur.stat("res_aborts"))
    idx = np.arange(N)
    with O.acc64():
        for f in check:
            ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, before[f], masses, idx, box)
            assert abs(float(r[f]) - ro) <= 1e-5, (f, f in bad, float(r[f]), ro)
            got = cur.get_positions(f)
            assert np.abs(got - want).max() <= 5e-5, (f, f in bad, float(np.abs(got - want).max()))
    plan.close(); ref.close(); cur.close()


@pytest.mark.parametrize("frac,resident", [(0.60, True), (0.30, False)])
def test_selection_fraction_decides_between_the_pass_and_the_two_passes(G, frac, resident):
    """the resident pass costs the same whatever share of the atoms is selected, the two passes shrink with the selection; the
    cross-over lies at ~40 % (profiles/r04_selection.json) and the default takes the pass from 45 %: a 60 % prefix runs
    k_fit_resident<.., V = false>, a 30 % prefix the two passes; both against the oracle"""
    nf = 24
    s_last = int(frac * N) - 2
    box, masses, cur, ref, ref_pos, plan = _c4(G, N, nf, sel=(0, s_last))
    prof, _ = _check(cur, plan, ref_pos, masses, np.arange(s_last + 1), box, [box] * nf, nf, [0, 7, nf - 1])
    assert (prof["k_fit_resident"][1] == 1) == resident and (prof["k_fit_pk"][1] > 0) == (not resident), prof
    plan.close(); ref.close(); cur.close()
