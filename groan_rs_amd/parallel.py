"""Frame-sharded map-reduce across GPUs: the multi-GPU form of System::traj_iter_map_reduce
(src/system/parallel.rs:208-481).

The reference clones the System per worker thread and gives worker n the frames n, n+T, n+2T, ...
(parallel.rs:424-448); every worker owns a `Data: ParallelTrajData`, and the results are merged with
`ParallelTrajData::reduce(Vec<Data>)` (parallel.rs:31-49,321).  Here a worker is one process per GPU
(torch.distributed rank; backend nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests); frames are
independent, so the only exchange is the final gather of the per-rank results -- no data-path collective.
"""
import ctypes as C

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


# ------------------------------------------------------------------------------------------------ behind the C ABI
class Pool:
    """gr_pool_*: the in-process form -- one worker thread + one device context per entry of `devices` (the per-thread System
    clones of parallel.rs:236), frames round-robin, shared error flag, per-frame results in frame order.

        pool = Pool([0, 0], n_atoms)                       # two workers on GPU 0 (or [0, 1, 2, 3]: one per GPU)
        for s in pool.systems: s.set_masses(m); s.group_create_from_ranges("Protein", [(0, 60)])
        results = pool.map(n_frames, body, width=3)        # body(system, worker, frame, out_row) -> None (raise on error)
    """
    BODY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.POINTER(C.c_float))

    def __init__(self, devices, n_atoms, n_slots=1):
        from . import _lib
        from .system import DeviceError, System
        self._lib = _lib.load()
        dev = (C.c_int * len(devices))(*[int(d) for d in devices])
        st = C.c_int(0)
        self._pool = self._lib.gr_pool_create(dev, len(devices), int(n_atoms), int(n_slots), C.byref(st))
        if not self._pool:
            raise DeviceError("PoolCreation", self._lib.gr_status_string(st.value).decode(), st.value)
        self.systems = [System._borrow(self._lib.gr_pool_ctx(self._pool, w), n_atoms, n_slots, devices[w], name="worker %d" % w)
                        for w in range(len(devices))]
        self.last_error = None

    def close(self):
        if getattr(self, "_pool", None):
            for s in self.systems:
                s.close()
            self._lib.gr_pool_destroy(self._pool)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    PROGRESS = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64)
    RUNNING, COMPLETED, FAILED = 0, 1, 2

    def map(self, n_frames, body, width, start=0, step=1, progress=None):
        """traj_iter_map_reduce's frame loop (parallel.rs:208-222,425-448): the frames start, start + step, ... below n_frames,
        worker w of T taking every T-th of them from the w-th on.  -> float32 [number of visited frames, width], row k = the k-th
        visited frame; raises the first failing frame's exception (the call as a whole fails, :288-321).
        progress(status, frame, frames_done): the ProgressPrinter -- RUNNING after every frame of worker 0, then once COMPLETED
        (frame = the last one any worker read) or FAILED (frame = the one that failed)."""
        if step < 1:
            raise ValueError("step must be >= 1 (ReadTrajError::InvalidStep)")
        n_visit = len(range(int(start), int(n_frames), int(step)))
        out = np.full((n_visit, max(width, 1)), np.nan, np.float32)
        errors = {}

        def trampoline(ctx, worker, frame, user, result):
            try:
                row = np.ctypeslib.as_array(result, shape=(width,)) if width else None    # width 0: the body returns nothing per frame
                body(self.systems[worker], worker, int(frame), row)
                return 0
            except Exception as e:   # the body's error: reported through the status, re-raised by map()
                errors[int(frame)] = e
                return getattr(e, "status", None) or 10
        cb = self.BODY(trampoline)
        pcb = self.PROGRESS((lambda user, status, frame, done: progress(int(status), int(frame), int(done))) if progress else 0)
        done, bad = C.c_uint64(0), C.c_uint64(0)
        st = self._lib.gr_pool_map_range(self._pool, int(start), int(n_frames), int(step), cb, None, int(width),
                                         out.ctypes.data_as(C.c_void_p) if width else None, pcb if progress else None, None, C.byref(done), C.byref(bad))
        self.frames_done = int(done.value)
        if st != 0:
            self.last_error = (int(bad.value), errors.get(int(bad.value)))
            raise errors.get(int(bad.value)) or RuntimeError("gr_pool_map_range failed at frame %d (status %d)" % (bad.value, st))
        return out if width else out[:, :0]


class Comm:
    """gr_comm_*: the multi-process form -- an RCCL communicator (ncclCommInitRank) for the final gather of per-frame results
    over xGMI and the shared error flag.  `unique_id()` on rank 0, hand the 128 bytes round, `Comm(device, rank, world, id)`."""

    @staticmethod
    def unique_id():
        from . import _lib
        buf = C.create_string_buffer(128)
        st = _lib.load().gr_comm_unique_id(buf)
        if st != 0:
            raise RuntimeError("gr_comm_unique_id: status %d (RCCL: %s)" % (st, _lib.load().gr_comm_library().decode()))
        return buf.raw

    def __init__(self, device, rank, world, unique_id):
        from . import _lib
        self._lib = _lib.load()
        st = C.c_int(0)
        self.rank, self.world = int(rank), int(world)
        self._comm = self._lib.gr_comm_create(int(device), self.rank, self.world, C.create_string_buffer(bytes(unique_id), 128), C.byref(st))
        if not self._comm:
            raise RuntimeError("gr_comm_create: status %d (RCCL: %s)" % (st.value, self._lib.gr_comm_library().decode()))

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.gr_comm_destroy(self._comm)
            self._comm = None

    def gather_per_frame(self, local_values, n_total):
        local = np.ascontiguousarray(local_values, dtype=np.float32)
        width = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
        mine = len(range(self.rank, int(n_total), self.world))         # rows the C side reads: this rank's frames rank, rank + G, ...
        if local.shape[0] < mine:
            raise ValueError("gather_per_frame: rank %d of %d holds %d of %d frames and must pass at least %d rows, got %d"
                             % (self.rank, self.world, mine, n_total, mine, local.shape[0]))
        out = np.zeros((n_total,) + tuple(local.shape[1:]), np.float32)
        st = self._lib.gr_comm_gather_per_frame(self._comm, local.ctypes.data_as(C.c_void_p), int(n_total), width, out.ctypes.data_as(C.c_void_p))
        if st != 0:
            raise RuntimeError("gr_comm_gather_per_frame: %s" % self._lib.gr_comm_last_error(self._comm).decode())
        return out

    def any_error(self, flag):
        r = C.c_int(0)
        st = self._lib.gr_comm_any_error(self._comm, int(bool(flag)), C.byref(r))
        if st != 0:
            raise RuntimeError("gr_comm_any_error: %s" % self._lib.gr_comm_last_error(self._comm).decode())
        return bool(r.value)
