"""A short run of tools/resident_fuzz.py: random system sizes (3 000 - 1 000 000 atoms), frames per call, numbers of frame streams,
selections, cells, boxes per frame and frames without a position -- the resident RMSD-fit pass against the two-pass path on the
same frames.  (profiles/r03_resident_fuzz.txt: 22 390 cases in fifteen minutes, no mismatch, no abort, no missed start.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_of_the_resident_pass_agree_with_the_two_passes(seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resident_fuzz.py"), "12", str(seed)], capture_output=True, text=True, timeout=300)
    tail = "\n".join(p.stdout.splitlines()[-5:])
    assert p.returncode == 0 and "mismatches 0" in p.stdout, tail + p.stderr[-2000:]
    assert "'aborts': 0" in p.stdout, tail
