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


def test_wrap_and_min_image_closed_forms_equal_the_reference_loops():
    """gr_math.h's branch-free wrap / min_image against the loops of src/structures/vector3d.rs:398-417,575-592, bit for bit
    where rounding decides (tiny negatives land exactly on L, exact multiples, neighbours of 0 / L / L/2)"""
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", d, "test_wrap"])
    p = subprocess.run([os.path.join(d, "test_wrap")], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0 and " 0 mismatches" in p.stdout


def test_minimum_image_table_is_complete_and_laid_out_in_pairs():
    """gr_box_setup's image table (gr_math.h) for the benchmark's cells, random GROMACS-reduced cells, cells on the limits and
    flat cells: every brick-reduced vector finds THE minimum image (fp64 search over 7 x 7 x 7 lattice vectors), the (t, t + a)
    pair layout of the packed searches keeps the order of the plain enumeration (tests/cpp/test_boxtable.cpp)"""
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", d, "test_boxtable"])
    p = subprocess.run([os.path.join(d, "test_boxtable")], capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr)
    assert p.returncode == 0 and "PASS" in p.stdout and " 0 failures" in p.stdout
