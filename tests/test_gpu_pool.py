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


def run(G, n_workers, example, fail_at=None):
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
        res = pool.map(nf, body, width=4)
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


def test_comm_single_rank_gather_and_flag(G):
    uid = G.Comm.unique_id()
    assert len(uid) == 128
    comm = G.Comm(0, 0, 1, uid)
    vals = (np.arange(37 * 4, dtype=np.float32) * 0.25).reshape(37, 4)
    got = comm.gather_per_frame(vals, 37)                                    # ncclAllGather over one rank: the identity
    assert np.array_equal(got, vals)
    assert comm.any_error(False) is False and comm.any_error(True) is True    # ncclAllReduce(MAX)
    comm.close()
