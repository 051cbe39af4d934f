"""Multi-GPU behind the C ABI: gr_pool_* -- worker threads + one context per device, frames round-robin
(src/system/parallel.rs:424-448), shared error flag, results in frame order -- with REAL kernels, two workers sharing the one
GPU of the test box; and gr_comm_* -- the RCCL communicator of the one-process-per-GPU form -- with a single rank (the box has
one GPU; RCCL refuses two ranks on one device).  Mirror of the reference's parallel tests (parallel.rs:932-1149): for every
worker count the per-frame results equal the single-worker results, here bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def run(G, n_workers, example, fail_at=None, start=0, step=1, progress=None):
    x = G.XtcFile(os.path.join(GOLD, "short_trajectory.xtc"))       # indexed, random access, thread-safe reads: every worker reads its own frames
    n, nf = x.n_atoms, x.n_frames
    masses = np.full(n, np.nan, np.float32); masses[:61] = example["protein_masses"]
    ref_pos, ref_box, _, _, _ = x.read_frame(0)
    pool = G.Pool([0] * n_workers, n)
    plans, refs = [], []
    for s in pool.systems:
        s.set_masses(masses); s.group_create_from_ranges("Protein", [(0, 60)])
        r = G.System(n, masses=masses, box=ref_box, positions=ref_pos); r.group_create_from_ranges("Protein", [(0, 60)])
        refs.append(r); plans.append(G.RMSDPlan(r, s, "Protein"))
    visited = [[] for _ in range(n_workers)]

    def body(system, worker, frame, out):
        if fail_at is not None and frame == fail_at:
            raise RuntimeError("body failed at frame %d" % frame)
        pos, box, _, _, _ = x.read_frame(frame)
        system.set_frame(pos, box)
        out[0:3] = system.group_get_com("Protein")
        out[3] = plans[worker].rmsd(0, 1)[0][0]
        visited[worker].append(frame)
    try:
        res = pool.map(nf, body, width=4, start=start, step=step, progress=progress)
        return res, visited, pool.frames_done
    finally:
        for p in plans: p.close()
        for r in refs: r.close()
        pool.close(); x.close()


def test_two_workers_equal_one_worker_bit_for_bit(G, example):
    one, v1, d1 = run(G, 1, example)
    two, v2, d2 = run(G, 2, example)
    three, v3, d3 = run(G, 3, example)
    assert d1 == d2 == d3 == 11 and np.isfinite(one).all()
    assert np.array_equal(one.view(np.uint32), two.view(np.uint32)) and np.array_equal(one.view(np.uint32), three.view(np.uint32))
    assert v1 == [list(range(11))]
    assert v2 == [[0, 2, 4, 6, 8, 10], [1, 3, 5, 7, 9]]                      # worker w: frames w, w + T, ... (parallel.rs:424-448)
    assert sorted(sum(v3, [])) == list(range(11))
    assert abs(float(one[0, 3])) <= 1e-4                                     # frame 0 against itself


def test_a_failing_body_fails_the_whole_call(G, example):
    with pytest.raises(RuntimeError, match="frame 5"):
        run(G, 2, example, fail_at=5)


def test_range_step_and_progress_printer(G, example):
    """traj_iter_map_reduce's start / step / progress_printer (parallel.rs:208-222): frames start, start + step, ...; worker w skips
    w * step of them, then advances by step * T (:425-448); the printer runs in the master worker and closes with the last frame
    ANY worker read (:300-317)"""
    full, _, _ = run(G, 1, example)
    log = []
    got, v, done = run(G, 2, example, start=1, step=3, progress=lambda st, fr, d: log.append((st, fr, d)))
    assert v == [[1, 7], [4, 10]] and done == 4                              # visited 1, 4, 7, 10: worker 0 -> k = 0, 2; worker 1 -> k = 1, 3
    assert np.array_equal(got.view(np.uint32), full[1::3].view(np.uint32))   # rows in visiting order
    assert [e[:2] for e in log if e[0] == 0] == [(0, 1), (0, 7)]             # RUNNING: the master worker's frames only
    assert log[-1][0] == 1 and log[-1][1] == 10 and log[-1][2] == 4          # COMPLETED with the last frame any worker read
    log = []
    with pytest.raises(RuntimeError, match="frame 4"):
        run(G, 2, example, fail_at=4, start=1, step=3, progress=lambda st, fr, d: log.append((st, fr, d)))
    assert log[-1][0] == 2 and log[-1][1] == 4                               # FAILED with the failing frame
    three, v3, _ = run(G, 3, example, start=2, step=2)
    assert v3 == [[2, 8], [4, 10], [6]] and np.array_equal(three.view(np.uint32), full[2::2].view(np.uint32))
    with pytest.raises(ValueError):
        run(G, 1, example, step=0)


def test_width_zero_bodies_and_short_gather_buffers_are_refused_cleanly(G, example):
    """a body that returns nothing per frame (width 0: the reduce happens in the caller's own Data) must not be handed a NULL row;
    a rank that passes fewer rows than its share of the frames is refused on the host instead of read past"""
    pool = G.Pool([0, 0], 100)
    seen = []
    out = pool.map(7, lambda system, worker, frame, row: seen.append((frame, row)), width=0)
    assert out.shape == (7, 0) and sorted(f for f, _ in seen) == list(range(7)) and all(r is None for _, r in seen)
    pool.close()
    comm = G.Comm(0, 0, 1, G.Comm.unique_id())
    with pytest.raises(ValueError):
        comm.gather_per_frame(np.zeros((3, 2), np.float32), 5)
    comm.close()


def test_comm_single_rank_gather_and_flag(G):
    uid = G.Comm.unique_id()
    assert len(uid) == 128
    comm = G.Comm(0, 0, 1, uid)
    vals = (np.arange(37 * 4, dtype=np.float32) * 0.25).reshape(37, 4)
    got = comm.gather_per_frame(vals, 37)                                    # ncclAllGather over one rank: the identity
    assert np.array_equal(got, vals)
    assert comm.any_error(False) is False and comm.any_error(True) is True    # ncclAllReduce(MAX)
    comm.close()
