"""Multi-rank path on CPU: world_size-2 gloo.  Frame sharding is the reference's round-robin
(src/system/parallel.rs:424-448: worker n takes frames n, n+T, ...); the only exchange is the final gather of
per-frame results (+ the shared error flag, parallel.rs:28,453-475).  Mirrors the reference's own parallel tests
(parallel.rs:932-1149: for every worker count, start and step the multiset of visited frames equals the serial one)."""
import os
import socket

import numpy as np
import pytest

from groan_rs_amd.parallel import AbortedByOtherRank, ParallelTrajData, gather_per_frame, interleave, shard_frames, traj_iter_map_reduce


def test_shard_frames_covers_exactly_like_the_reference():
    for n_frames in (0, 1, 11, 100, 1001):
        for start in (0, 3):
            for step in (1, 2, 5):
                serial = list(range(start, n_frames, step))
                for world in list(range(1, 17)) + [21]:
                    shards = [shard_frames(n_frames, r, world, start, step) for r in range(world)]
                    assert sorted(sum(shards, [])) == serial
                    for r, sh in enumerate(shards):          # worker r: frames start + (r + k*world)*step
                        assert sh == serial[r::world]


def test_interleave_restores_frame_order():
    for world in (1, 2, 3, 8):
        for n in (1, 7, 64, 1001):
            full = np.arange(n, dtype=np.float32) * 0.5
            per = (n + world - 1) // world
            shards = []
            for r in range(world):
                s = np.zeros(per, np.float32); loc = full[r::world]; s[: loc.size] = loc
                shards.append(s)
            assert np.array_equal(interleave(shards, n), full)


class _FakeSystem:
    def __init__(self):
        self.frames = []

    def set_frame(self, pos, box, slot=0, step=None, time=None):
        self.frames.append(step)


class _Data(ParallelTrajData):
    def __init__(self):
        self.seen, self.rank = [], None

    def initialize(self, thread_id):
        self.rank = thread_id

    @staticmethod
    def reduce(data):
        out = _Data()
        for d in data:
            out.seen += d.seen
        out.seen.sort()
        return out


def _worker(rank, world, port, n_frames, fail_at, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames_of = lambda idxs: ((None, None, i, float(i)) for i in idxs)

        def body(system, data):
            f = system.frames[-1]
            if fail_at is not None and f == fail_at:
                raise RuntimeError("body failed at frame %d" % f)
            data.seen.append(f)

        err = None
        try:
            data = traj_iter_map_reduce(lambda r: _FakeSystem(), frames_of, n_frames, body, _Data(), rank, world, dist)
        except AbortedByOtherRank as e:
            err, data = "aborted: " + str(e), None
        except RuntimeError as e:
            err, data = str(e), None
        # final gather of a per-frame scalar (what the RMSD bench gathers over RCCL)
        mine = np.array([float(i) * 2.0 for i in shard_frames(n_frames, rank, world)], np.float32)
        full = gather_per_frame(mine, n_frames, dist=dist)
        q.put((rank, None if data is None else (data.rank, data.seen), err, full.tolist()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("fail_at", [None, 35, 100])
def test_world_size_2_gloo(fail_at):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_frames, world, port = 101, 2, _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, fail_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [float(i) * 2.0 for i in range(n_frames)]
    for rank, data, err, full in res:
        assert full == want                                   # every rank holds the gathered, order-restored result
    if fail_at is None:
        assert all(err is None for _, _, err, _ in res)
        merged = _Data.reduce([type("D", (), {"seen": d[1]})() for _, d, _, _ in res])
        assert merged.seen == list(range(n_frames))           # multiset of visited frames == serial
        assert [d[0] for _, d, _, _ in res] == [0, 1]         # initialize(thread_id)
    else:
        # the call fails on EVERY rank (the reference returns Err for the whole map-reduce, parallel.rs:288-321): the rank whose
        # body failed re-raises its error, the other one raises AbortedByOtherRank -- nobody returns partial data as success.
        # fail_at = 100 is the LAST frame (rank 0's last round): only the final flag reduction after the loop can share it.
        errs = [err for _, _, err, _ in res]
        assert all(e is not None for e in errs), errs
        assert sum(("frame %d" % fail_at) in e for e in errs) == 1
        assert sum(e.startswith("aborted: ") for e in errs) == 1
        assert all(d is None for _, d, _, _ in res)


def test_abi_deinterleave_matches_the_python_gather():
    """gr_shard_deinterleave (the host half of gr_comm_gather_per_frame; no device, no RCCL) == interleave()"""
    import ctypes as C
    from groan_rs_amd import _lib
    lib = _lib.load()
    for world in (1, 2, 3, 8):
        for n in (1, 7, 64, 1001):
            for width in (1, 4):
                full = (np.arange(n * width, dtype=np.float32) * 0.5).reshape(n, width)
                per = (n + world - 1) // world
                shards = np.zeros((world, per, width), np.float32)
                for r in range(world):
                    loc = full[r::world]; shards[r, : loc.shape[0]] = loc
                out = np.zeros((n, width), np.float32)
                lib.gr_shard_deinterleave(shards.ctypes.data_as(C.c_void_p), world, n, width, out.ctypes.data_as(C.c_void_p))
                assert np.array_equal(out, full) and np.array_equal(out, interleave([s for s in shards], n))
