"""The benchmark's multi-rank path rehearsed on ONE GPU (the test box has one): `bench.py --gpus 2` launched exactly as the driver
launches it (python -m torch.distributed.run, one process per rank, rendezvous on 127.0.0.1) with GROAN_DIST_BACKEND=gloo -- RCCL
refuses two ranks on one device, so the final gather goes over gloo; everything else (frames sharded round-robin over the ranks as
src/system/parallel.rs:424-448, one context per rank, the barrier + max-over-ranks timing, the gather and its de-interleave) is the
code the 8-GPU run executes.  Children are fresh processes started with subprocess (nothing that has touched the GPU is re-executed).
Checked: every rank's frames arrive (gathered_frames = ranks x steps x frames), in global frame order, and are BIT-IDENTICAL to
the same frames processed by a single rank; with the resident pass forced, the two ranks' launches share the device (each fits
beside the other, or one takes the two-pass path for a segment) and neither aborts."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(tmp_path, tag, n_ranks, frames_per_step, extra):
    dump = str(tmp_path / ("rmsd_%s_%d.npy" % (tag, n_ranks)))
    args = ["bench.py", "--gpus", str(n_ranks), "--atoms", "200000", "--steps", "3", "--warmup", "1", "--frames-per-step", str(frames_per_step),
            "--no-cpu-baseline", "--dump-rmsd", dump] + extra
    if n_ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:
        cmd = [sys.executable] + args
    env = dict(os.environ, GROAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    return json.loads(lines[0]), np.load(dump)


@pytest.mark.parametrize("tag,extra", [("default", []), ("resident", ["--tune", "resident=2"])])
def test_two_ranks_on_one_gpu_equal_one_rank_bit_for_bit(tmp_path, tag, extra):
    B = 256
    two, r2 = _bench(tmp_path, tag, 2, B, extra)
    one, r1 = _bench(tmp_path, tag, 1, 2 * B, extra)
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["scaling"] == "weak"
    assert two["config"]["gathered_frames"] == 2 * 3 * B == r2.shape[0]
    assert len(two["config"]["per_rank_frames_per_s"]) == 2 and two["value"] > 0
    # global frame g was processed by rank g % 2 as its local frame g // 2; the single rank processed the same frames in order
    assert r1.shape == r2.shape and np.isfinite(r2).all()
    misses = sum(st["res_handshake_misses"] for st in two["config"]["per_rank_resident"] + one["config"]["per_rank_resident"])
    if misses == 0:
        assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32))
    else:      # a rank found the chip full at a launch and ran that segment on the two-pass path: same results up to the order of the sums
        assert np.abs(r1 - r2).max() <= 2e-6
    for st in two["config"]["per_rank_resident"]:
        assert st["res_aborts"] == 0 and st["res_redone_frames"] == 0, two["config"]["per_rank_resident"]
    if tag == "resident":
        # each rank asked for the resident pass: its launches ran, or found the device taken / the chip full and fell back cleanly
        assert all(st["res_launches"] + st["res_handshake_misses"] >= 0 for st in two["config"]["per_rank_resident"])
        assert one["config"]["per_rank_resident"][0]["res_launches"] >= 3
        assert sum(st["res_launches"] for st in two["config"]["per_rank_resident"]) >= 1, two["config"]["per_rank_resident"]


def test_bench_gpus_2_without_a_launcher_starts_two_ranks(tmp_path):
    """`python bench.py --gpus 2` (no torch.distributed.run in front, WORLD_SIZE unset) must MEAN two ranks: the parent starts them as
    fresh child processes before it touches the GPU and relays rank 0's line (--gpus used to be parsed and ignored)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GROAN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--atoms", "200000", "--steps", "2", "--warmup", "1", "--frames-per-step", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["gathered_frames"] == 2 * 2 * 64 and len(out["config"]["per_rank_frames_per_s"]) == 2
