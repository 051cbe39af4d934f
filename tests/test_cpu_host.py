"""CPU-only checks of the product's host side: the C-ABI library loads, exports every symbol that
include/groan_hip.h declares, its pure-host AtomContainer functions are bit-exact against the reference's
known answers, and -- with no GPU present -- context creation fails loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "groan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", txt)))


def test_abi_exports_every_declared_symbol(G):
    lib = G._lib.load()
    names = declared_functions()
    assert len(names) >= 50
    for n in names:
        assert hasattr(lib, n), "libgroan_hip.so does not export %s" % n
        assert n in G._lib.SIGNATURES, "%s has no ctypes signature" % n
    assert sorted(G._lib.SIGNATURES) == names
    assert lib.gr_version().startswith(b"groan_hip")
    assert lib.gr_status_string(6) == b"atom has undefined position"


def test_no_device_fails_loudly(G):
    lib = G._lib.load()
    n = C.c_int(-1)
    st = lib.gr_device_count(C.byref(n))
    if st == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(G.DeviceError) as e:
        G.System(10)
    assert e.value.status == G._lib.E_NO_DEVICE


def _b(c):
    return c.blocks


def test_container_goldens_through_the_abi(G):
    """src/structures/container.rs:517-925 (same vectors the oracle is pinned with)"""
    AC = G.AtomContainer
    assert _b(AC.from_indices([6, 2, 13, 1, 10, 8, 3, 12, 7, 14, 15], 16)) == [(1, 3), (6, 8), (10, 10), (12, 15)]
    assert _b(AC.from_indices([], 16)) == []
    dup = [1, 6, 3, 2, 13, 1, 10, 8, 3, 12, 7, 14, 15, 10]
    assert _b(AC.from_indices(dup, 16)) == [(1, 3), (6, 8), (10, 10), (12, 15)]
    assert _b(AC.from_indices(dup, 15)) == [(1, 3), (6, 8), (10, 10), (12, 14)]
    cx = [11, 1, 2, 3, 20, 5, 0, 5, 4, 18, 6, 19, 1, 13, 20, 27]
    c1 = AC.from_indices(cx, 20)
    assert _b(c1) == [(0, 6), (11, 11), (13, 13), (18, 19)]
    assert c1.get_n_atoms() == 11 and list(c1) == [0, 1, 2, 3, 4, 5, 6, 11, 13, 18, 19]
    assert c1.first() == 0 and c1.last() == 19 and c1.isin(5) and not c1.isin(12) and not c1.isin(73)
    assert _b(AC.from_ranges([(64, 128), (5, 32), (1, 25), (129, 133), (133, 200), (35, 78), (10, 15), (1033, 1055)], 1028)) == [(1, 32), (35, 200)]
    assert _b(AC.from_ranges([(1, 25), (0, 1), (0, 0), (0, 34)], 1028)) == [(0, 34)]
    assert _b(AC.from_ranges([(32, 25), (14, 17)], 1028)) == [(14, 17)]
    assert _b(AC.from_ranges([(543, 1020), (1000, 1432)], 1028)) == [(543, 1027)]
    assert _b(AC.from_ranges([(0, 43), (1006, 1432)], 1028)) == [(0, 43), (1006, 1027)]
    c2 = AC.from_ranges([(10, 15), (17, 25), (11, 11), (7, 3), (9, 10), (15, 15), (16, 18), (2, 5), (10, 15)], 20)
    assert _b(c2) == [(2, 5), (9, 19)]
    assert _b(AC.union(c1, c2)) == [(0, 6), (9, 19)]
    c3 = AC.from_indices([13, 1, 2, 7, 5, 19, 21, 1, 9, 10, 11], 15)
    assert _b(AC.intersection(c1, c3)) == [(1, 2), (5, 5), (11, 11), (13, 13)] == _b(AC.intersection(c3, c1))
    assert _b(AC.intersection(c1, AC())) == [] and _b(AC.intersection(AC(), c1)) == []


def test_container_random_against_oracle(G):
    AC = G.AtomContainer
    rng = np.random.default_rng(3)

    def bl(x):
        return [(int(a), int(b)) for a, b in np.asarray(x).tolist()]

    for _ in range(300):
        n_atoms = int(rng.integers(1, 300))
        idx = rng.integers(0, n_atoms + 30, size=int(rng.integers(0, 80))).tolist()
        if idx and min(idx) >= n_atoms:
            idx.append(0)   # the reference never range-checks the smallest index (container.rs:66): keep it valid
        r = [(int(a), int(b)) for a, b in rng.integers(0, n_atoms + 30, size=(int(rng.integers(0, 25)), 2))]
        c1, o1 = AC.from_indices(idx, n_atoms), O.container_from_indices(idx, n_atoms)
        c2, o2 = AC.from_ranges(r, n_atoms), (O.container_from_ranges(r, n_atoms) if r else np.zeros((0, 2), np.uint64))
        assert _b(c1) == bl(o1) and _b(c2) == bl(o2)
        assert c1.get_n_atoms() == O.container_expand(o1).size and list(c1) == O.container_expand(o1).tolist()
        assert _b(AC.union(c1, c2)) == bl(O.container_union(o1, o2))
        assert _b(AC.intersection(c1, c2)) == bl(O.container_intersection(o1, o2))
        for q in rng.integers(0, n_atoms + 30, size=5):
            assert c1.isin(int(q)) == O.container_isin(o1, int(q))


def test_dimension_mirror(G):
    D = G.Dimension   # src/structures/dimension.rs
    assert [d.is_x() for d in D] == [False, True, False, False, True, True, False, True]
    assert [d.is_y() for d in D] == [False, False, True, False, True, False, True, True]
    assert [d.is_z() for d in D] == [False, False, False, True, False, True, True, True]


def test_container_validate_refuses_what_from_indices_lets_through(G):
    """AtomContainer::from_indices never range-checks its smallest index (container.rs:66): [n] -> block (n, n),
    [n, n + 1] -> block (n, n - 1).  The reference panics on the first access through such a container; here no kernel
    bounds-checks a selection, so gr_container_validate (applied by every entry point that takes a selection) must refuse them."""
    lib = G._lib.load()
    n = 100

    def validate(indices):
        c = G.AtomContainer.from_indices(indices, n)
        s = np.array([b[0] for b in c.blocks], np.uint64); e = np.array([b[1] for b in c.blocks], np.uint64)
        bad = C.c_uint64(0)
        st = lib.gr_container_validate(s.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), len(c.blocks), n, C.byref(bad))
        return c.blocks, st, bad.value

    assert validate([105]) == ([(105, 105)], G._lib.E_OUT_OF_RANGE, 105)            # the quirk itself is kept bit-exact ...
    assert validate([100, 101]) == ([(100, 99)], G._lib.E_OUT_OF_RANGE, 100)        # ... and caught before any kernel sees it
    assert validate([100])[1:] == (G._lib.E_OUT_OF_RANGE, 100)                       # a 1-based last-atom index
    assert validate([256 + 3])[1:] == (G._lib.E_OUT_OF_RANGE, 259)                   # beyond the padded slot as well
    assert validate([5, 6, 7, 250])[1] == G._lib.OK                                  # out-of-range tail: clamped by the reference's scan
    assert validate([0, 99])[1] == G._lib.OK and validate([])[1] == G._lib.OK


def test_comm_without_rccl_reports_no_device_instead_of_crashing(tmp_path):
    """a host whose RCCL cannot be loaded: every gr_comm_* entry point returns GR_E_NO_DEVICE and gr_comm_library() carries the
    loader's message (the first version of the loader called dlerror() twice and dereferenced the NULL of the second call).  Run
    in a fresh process: the library is resolved once per process."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "sys.path.insert(0, %r)\n"
        "import groan_rs_amd as G\n"
        "lib = G._lib.load()\n"
        "assert lib.gr_comm_set_library(%r.encode()) == 0\n"
        "buf = C.create_string_buffer(128)\n"
        "st = C.c_int(0)\n"
        "assert lib.gr_comm_unique_id(buf) == 13, 'unique_id'\n"
        "assert not lib.gr_comm_create(0, 0, 1, buf, C.byref(st)) and st.value == 13, 'create'\n"
        "msg = lib.gr_comm_library().decode()\n"
        "assert msg.startswith('RCCL not found: ') and 'no_such_rccl' in msg, msg\n"
        "assert lib.gr_comm_set_library(b'librccl.so.1') == 10, 'an override after the library was resolved must be refused'\n"
        "print('ok')\n") % (ROOT, str(tmp_path / "no_such_rccl.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-300:], r.stderr[-600:])


def test_tuning_and_stat_keys_of_the_python_mirror_are_the_header_s(G):
    """gr_ctx_set_tuning / gr_ctx_stat take integer keys: the mirror's name -> key tables must be the header's enumerations, key by key
    (a key added to one and not the other would silently tune something else)"""
    txt = open(os.path.join(ROOT, "include", "groan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    tune = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"\bGR_TUNE_([A-Z0-9_]+)\s*=\s*(\d+)", txt)}
    stat = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"\bGR_STAT_([A-Z0-9_]+)\s*=\s*(\d+)", txt)}
    assert len(tune) >= 18 and len(stat) >= 12
    assert G.System.TUNE == tune, (sorted(set(G.System.TUNE.items()) ^ set(tune.items())))
    assert G.System.STAT == stat, (sorted(set(G.System.STAT.items()) ^ set(stat.items())))
