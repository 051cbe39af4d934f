"""Runs the C++ host-mirror test program (tests/cpp/test_host.cpp over include/groan_hip.hpp) on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(tmp_path, example, short_traj):
    exe = os.path.join(ROOT, "tests", "cpp", "test_host")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    n = short_traj["keep"].size
    m = np.full(n, np.nan, np.float32); m[:61] = example["protein_masses"]
    short_traj["gro_keep"].astype("<f4").tofile(tmp_path / "gro_keep.f32")
    short_traj["frames"].astype("<f4").tofile(tmp_path / "frames.f32")
    short_traj["boxes9"].astype("<f4").tofile(tmp_path / "boxes.f32")
    example["box9"].astype("<f4").tofile(tmp_path / "box0.f32")
    m.astype("<f4").tofile(tmp_path / "masses.f32")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "groan_rs_amd") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe, str(tmp_path), os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, env=env, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all passed" in p.stdout
