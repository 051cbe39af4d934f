"""Frame-sharded map-reduce across GPUs: the multi-GPU form of System::traj_iter_map_reduce
(src/system/parallel.rs:208-481).

The reference clones the System per worker thread and gives worker n the frames n, n+T, n+2T, ...
(parallel.rs:424-448); every worker owns a `Data: ParallelTrajData`, and the results are merged with
`ParallelTrajData::reduce(Vec<Data>)` (parallel.rs:31-49,321).  Here a worker is one process per GPU
(torch.distributed rank; backend nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests); frames are
independent, so the only exchange is the final gather of the per-rank results -- no data-path collective.
"""
import numpy as np


class ParallelTrajData:
    """parallel.rs:31-49"""

    def initialize(self, thread_id):
        pass

    @staticmethod
    def reduce(data):
        raise NotImplementedError


def shard_frames(n_frames, rank, world, start=0, step=1):
    """Frame indices of worker `rank`: start + (rank + k*world)*step  (parallel.rs:424-448)."""
    return list(range(start + rank * step, n_frames, step * world))


def interleave(shards, n_total):
    """Inverse of the round-robin sharding: out[f] = shards[f % G][f // G] (order restore after the gather)."""
    world = len(shards)
    first = np.asarray(shards[0])
    out = np.zeros((n_total,) + first.shape[1:], dtype=first.dtype)
    for r, sh in enumerate(shards):
        sh = np.asarray(sh)
        cnt = len(range(r, n_total, world))
        out[r::world] = sh[:cnt]
    return out


def gather_per_frame(local_values, n_total, dist=None, device=None):
    """Final gather of per-frame scalars/vectors (4-40 bytes per frame) from all ranks, restored to frame order.

    local_values: array [n_local, ...] for frames rank, rank+G, ...  Every rank receives the full result.
    With torch.distributed initialised this is ONE all_gather of ceil(n_total/G) rows per rank
    (RCCL over xGMI when the backend is nccl); without it (single process) it is the identity."""
    local = np.ascontiguousarray(local_values)
    if dist is None or not dist.is_initialized():
        return interleave([local], n_total)
    import torch
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    pad = np.zeros((per,) + local.shape[1:], dtype=local.dtype)
    pad[: local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return interleave([o.cpu().numpy() for o in outs], n_total)


class AbortedByOtherRank(RuntimeError):
    """Raised on the ranks that stopped because ANOTHER rank's body failed: the call as a whole failed
    (the reference returns the failing worker's Err for the whole traj_iter_map_reduce, parallel.rs:288-321),
    so no rank may hand its partial Data on as a success."""


def _flag_any(dist, world, device, raised):
    """all_reduce(MAX) of the shared error flag (the AtomicBool of parallel.rs:28,230).  The tensor lives where the
    backend can reduce it: on `device` for nccl / RCCL (CPU tensors have no nccl backend), on the CPU for gloo."""
    if dist is None or not dist.is_initialized() or world <= 1:
        return raised
    import torch
    dev = device
    if dev is None and dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    flag = torch.tensor([1 if raised else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(int(flag.item()))


def traj_iter_map_reduce(make_system, frames_of, n_frames, body, init_data, rank=0, world=1, dist=None,
                         start=0, step=1, error_flag_freq=10, device=None):
    """System::traj_iter_map_reduce for one worker.

    make_system(rank) -> System clone living on this worker's GPU (parallel.rs:236)
    frames_of(indices) -> iterable of frames (positions, box[, step, time]) for those frame indices
    body(system, data) -> None or raises; data = copy of init_data, initialised with the worker id
    device: torch device of the flag tensor (required form for backend nccl: the rank's GPU; None = CPU / current GPU)
    Returns the rank-local Data; callers reduce with ParallelTrajData.reduce after gathering.
    An error on any worker stops the others at their next check, every ERROR_FLAG_FREQ frames
    (parallel.rs:28,453-475): with torch.distributed the flag is a 1-int all_reduce(MAX); one more reduction after
    the last frame shares an error raised after the last periodic check.  The failing rank re-raises its error, every
    other rank raises AbortedByOtherRank: like the reference's Err for the whole call, nobody returns partial data.
    """
    import copy
    data = copy.deepcopy(init_data)
    data.initialize(rank)
    system = make_system(rank)
    mine = shard_frames(n_frames, rank, world, start, step)
    err = None
    # every rank runs the same number of flag checks so the collective stays matched
    n_rounds = (len(shard_frames(n_frames, 0, world, start, step)) + error_flag_freq - 1) // error_flag_freq
    it = iter(frames_of(mine))
    done = False
    aborted = False
    n_done = 0
    for _round in range(n_rounds):
        if _flag_any(dist, world, device, err is not None):
            aborted = True
            break
        for _ in range(error_flag_freq):
            if done or err is not None:
                break
            try:
                fr = next(it)
            except StopIteration:
                done = True
                break
            try:
                system.set_frame(fr[0], fr[1], slot=0, step=fr[2] if len(fr) > 2 else None, time=fr[3] if len(fr) > 3 else None)
                body(system, data)
                n_done += 1
            except Exception as e:   # first error wins (parallel.rs:468-471)
                err = e
    if not aborted:                  # the final check: an error of the last round is shared, too
        aborted = _flag_any(dist, world, device, err is not None)
    if err is not None:
        raise err
    if aborted:
        raise AbortedByOtherRank("another rank's frame body failed; this rank stopped after %d of its %d frames" % (n_done, len(mine)))
    return data
