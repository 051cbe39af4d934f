"""GPU parity of the persistent LDS-resident RMSD-fit kernel (groan_rs_amd/csrc/gr_persist.h) against the
accumulate -> finalize -> fit kernels and the oracle (reference: System::calc_rmsd_and_fit, src/system/rmsd.rs:113-166,
RMSDConverterAnalyzer::convert_analyze :229-251).  Both GPU paths evaluate the same per-atom arithmetic; only the
summation tree and the place the frame waits for its rotation (LDS instead of HBM) differ."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
SEED = 77


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def blob(G, n, box, nf, noise=0.04, group=None):
    masses = np.array([1.008, 12.011, 14.007, 15.999, 30.974], np.float32)[np.arange(n) % 5]
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.synth_reference(nf, box, 0.18 * float(min(box[:3])), SEED)
    cur.synth_frames(nf, 0, nf, 3, noise, SEED)
    ref_pos = cur.get_positions(nf)
    ref = G.System(n, masses=masses, box=box, positions=ref_pos)
    if group is not None:
        for s in (ref, cur):
            s.group_create_from_ranges("sel", [group])
    return ref, cur, masses, ref_pos


def run_both(G, ref, cur, nf, group, frames, boxes=None):
    """-> (rmsd, status, fitted) of the persistent kernel and of the three-kernel path on identical inputs"""
    out = []
    for persistent in (True, False):
        for f in range(nf):
            cur.set_frame(frames[f], "keep" if boxes is None else boxes[f], slot=f)
        cur.set_persistent(2 if persistent else 0)
        plan = G.RMSDPlan(ref, cur, group)
        r, st = plan.rmsd_fit(0, nf, raise_on_error=False)
        assert plan.last_persistent() == persistent
        out.append((r, st, [cur.get_positions(f) for f in range(nf)], plan.last_fallbacks()))
        plan.close()
    cur.set_persistent(0)
    return out


@pytest.mark.parametrize("n,nf,group,angles", [
    (100_003, 7, None, [60.0, 60.0, 90.0]),            # ragged tail tile, ring wraps twice (depth 3)
    (65_536, 2, None, [90.0, 90.0, 90.0]),             # smallest system / shortest batch the kernel takes
    (300_000, 33, (1234, 250_000), [75.0, 80.0, 70.0]),  # sub-range selection: edge groups, atoms outside are still transformed
    (131_072, 5, (4, 131_067), [70.53, 109.47, 70.53]),
])
def test_persistent_matches_three_kernel_path_and_oracle(G, n, nf, group, angles):
    box = O.box_from_lengths_angles([9.0, 9.0, 9.0] if angles[0] != 75.0 else [9.5, 9.0, 8.5], angles)
    ref, cur, masses, ref_pos = blob(G, n, box, nf, group=group)
    name = "all" if group is None else "sel"
    frames = [cur.get_positions(f) for f in range(nf)]
    (rp, sp, xp, fbp), (rk, sk, xk, fbk) = run_both(G, ref, cur, nf, name, frames)
    assert (sp == 0).all() and (sk == 0).all() and fbp == 0 and fbk == 0
    assert np.abs(rp - rk).max() <= 2e-6
    for f in range(nf):
        assert np.abs(xp[f] - xk[f]).max() <= 2e-5
    idx = np.arange(n) if group is None else np.arange(group[0], group[1] + 1)
    with O.acc64():
        for f in (0, nf - 1):
            ro, want = O.calc_rmsd_and_fit(ref_pos, masses, idx, box, frames[f], masses, idx, box)
            assert abs(rp[f] - ro) <= TOL
            np.testing.assert_allclose(xp[f], want, atol=3e-5, rtol=0)
    ref.close(); cur.close()


def test_persistent_frames_that_need_the_exact_path_or_fail(G):
    """inside one batch: a frame whose group spans more than half the box (image proof fails -> multi-pass path), a
    frame with a NaN position (error names the atom), a frame with its own different box -- the neighbours are untouched"""
    n, nf = 80_000, 6
    box = np.array([8.0, 7.0, 6.0, 0, 0, 0, 0, 0, 0], np.float32)
    rng = np.random.default_rng(5)
    base = rng.normal(0, 0.3, (n, 3))
    m = np.array([1.008, 12.011, 15.999], np.float32)[np.arange(n) % 3]
    ref_pos = O.wrap_atoms((base + [4.0, 3.5, 3.0]).astype(np.float32), np.arange(n), box)
    ref = G.System(n, masses=m, box=box, positions=ref_pos)
    cur = G.System(n, masses=m, n_slots=nf)
    frames, boxes = [], []
    for f in range(nf):
        x = ref_pos + rng.normal(0, 0.02, (n, 3)).astype(np.float32)
        x = O.translate(x, np.arange(n), [1.1 * f, -0.7 * f, 0.4 * f], box)
        frames.append(x); boxes.append(box.copy())
    wide = base.copy(); wide[: n // 3, 0] += 4.6
    frames[2] = O.wrap_atoms((wide + [1.0, 3.5, 3.0]).astype(np.float32), np.arange(n), box)
    frames[4] = frames[4].copy(); frames[4][54_321, 0] = np.nan   # Option<Vector3D>::None travels as NaN in x
    boxes[5] = np.array([9.0, 7.5, 6.5, 0, 0, 0, 0, 0, 0], np.float32)
    (rp, sp, xp, fbp), (rk, sk, xk, fbk) = run_both(G, ref, cur, nf, "all", frames, boxes)
    assert np.array_equal(sp, sk) and fbp == fbk and fbp >= 1
    assert sp[4] != 0 and all(sp[f] == 0 for f in (0, 1, 2, 3, 5))
    ok = [0, 1, 2, 3, 5]
    assert np.abs(rp[ok] - rk[ok]).max() <= 2e-6
    for f in ok:
        assert np.abs(xp[f] - xk[f]).max() <= 2e-5
    got4 = xp[4]
    same = ~np.isnan(frames[4])
    assert np.array_equal(got4[same], frames[4][same])                 # the failed frame is left as it was
    with O.acc64():
        for f in (2, 5):
            ro, want = O.calc_rmsd_and_fit(ref_pos, m, np.arange(n), box, frames[f], m, np.arange(n), boxes[f])
            assert abs(rp[f] - ro) <= TOL
            np.testing.assert_allclose(xp[f], want, atol=3e-5, rtol=0)
    cur.set_frame(frames[4], boxes[4], slot=0)
    plan = G.RMSDPlan(ref, cur, "all")
    with pytest.raises(G.RMSDError) as e:
        plan.rmsd_fit(0, 2)
    assert e.value.variant == "InvalidPosition"
    ref.close(); cur.close()


def test_persistent_back_to_back_batches_and_begin_end(G):
    """sync words are re-armed per launch; begin/end leaves the stream free for the next uploads"""
    n, nf = 70_000, 9
    box = O.box_from_lengths_angles([8.0, 8.0, 8.0], [60.0, 60.0, 90.0])
    ref, cur, masses, ref_pos = blob(G, n, box, nf)
    cur.set_persistent(2)
    plan = G.RMSDPlan(ref, cur, "all")
    first, _ = plan.rmsd_fit(0, nf)
    assert plan.last_persistent()
    for rep in range(3):
        cur.synth_frames(nf, 0, nf, 3, 0.04, SEED)
        plan.begin(0, nf, True)
        r, st = plan.end()
        assert (st == 0).all() and np.array_equal(r, first)          # same inputs, same kernel -> same bits
    again, _ = plan.rmsd_fit(0, nf)                                    # already fitted: rmsd unchanged, rotation ~ identity
    assert np.abs(again - first).max() <= TOL
    ref.close(); cur.close()
