"""bench.py's launch convention, CPU side: a rank count that disagrees with the launcher's is an error, not a silent single-rank run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_that_disagrees_with_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_gpus_flag_defaults_agree_with_a_plain_run():
    """`python bench.py` (no flags, no launcher) stays a one-rank run: it must get past the rank check (and then fail loudly, here,
    because there is no GPU: the product has no CPU path)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--atoms", "1000", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert "WORLD_SIZE" not in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
