"""Host-side checks of the 3x3 rotation algebra the GPU finalize step uses (groan_rs_amd/csrc/gr_rotation.h): the Newton
polar fast path against the Jacobi / Kabsch restatement of src/system/rmsd.rs:573-583 (tests/cpp/test_rotation.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_polar_fast_path_agrees_with_kabsch_jacobi():
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", d, "test_rotation"])
    p = subprocess.run([os.path.join(d, "test_rotation")], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0 and "PASS" in p.stdout
