import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(HERE, "golden", name))


def ulps(a, b):
    a = np.float32(a); b = np.float32(b)
    ia = int(np.frombuffer(a.tobytes(), np.int32)[0]); ib = int(np.frombuffer(b.tobytes(), np.int32)[0])
    if (ia < 0) != (ib < 0):
        return 1 << 31
    return abs(ia - ib)


def approx_f32(a, b, epsilon=None, ulps_margin=4):
    """float_cmp::approx_eq!(f32, a, b[, epsilon = e]) -- default margin: f32::EPSILON or 4 ulps."""
    a = float(np.float32(a)); b = float(np.float32(b))
    eps = float(np.finfo(np.float32).eps) if epsilon is None else epsilon
    if a == b or abs(a - b) <= eps:
        return True
    return ulps(a, b) <= ulps_margin


def assert_approx(a, b, epsilon=None, msg=""):
    assert approx_f32(a, b, epsilon), "%r != %r (eps=%r) %s" % (float(a), float(b), epsilon, msg)


@pytest.fixture(scope="session")
def example():
    return golden("example.npz")


@pytest.fixture(scope="session")
def short_traj():
    return golden("short_traj.npz")


@pytest.fixture(scope="session")
def aa():
    return golden("aa_peptide.npz")


@pytest.fixture(scope="session")
def tric_small():
    return golden("tric_small.npz")
