"""ctypes binding of oracle/liboracle.so -- the CPU parity oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)

OK, E_NO_BOX, E_NOT_ORTHOGONAL, E_ZERO_BOX, E_EMPTY_GROUP, E_INCONSISTENT_GROUP, E_NO_POSITION, E_NO_MASS = range(8)
DIM = {"none": 0, "x": 1, "y": 2, "z": 3, "xy": 4, "xz": 5, "yz": 6, "xyz": 7}


_PATH = None


def use_library(path):
    """load another build of the same oracle (bench.py's CPU-baseline child: the -march=native variant built on that host)"""
    global _LIB, _PATH
    _LIB, _PATH = None, path


def lib():
    global _LIB
    if _LIB is None:
        path = _PATH or os.path.join(ROOT, "oracle", "liboracle.so")
        src = os.path.join(ROOT, "oracle", "groan_oracle.c")
        if _PATH is None and (not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"],
                                  stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.go_floor_mod.restype = C.c_float
        L.go_floor_mod.argtypes = [C.c_float, C.c_float]
        L.go_wrap_coordinate.restype = C.c_float
        L.go_wrap_coordinate.argtypes = [C.c_float, C.c_float]
        L.go_min_image.restype = C.c_float
        L.go_min_image.argtypes = [C.c_float, C.c_float]
        L.go_distance.restype = C.c_float
        L.go_distance_naive.restype = C.c_float
        L.go_container_n_atoms.restype = C.c_uint64
        for name in ("go_container_from_indices", "go_container_from_ranges", "go_container_union",
                     "go_container_intersection", "go_container_expand"):
            getattr(L, name).restype = C.c_size_t
        L.go_baseline_rmsd_fit.restype = C.c_double
        _LIB = L
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _box(b):
    if b is None:
        return None
    b = _f(b).ravel()
    if b.size == 3:
        b = np.concatenate([b, np.zeros(6, np.float32)])
    assert b.size == 9
    return np.ascontiguousarray(b)


class OracleError(Exception):
    def __init__(self, status, index=None, counts=None):
        super().__init__("oracle status %d index %s" % (status, index))
        self.status, self.index, self.counts = status, index, counts


def set_strict_orthogonal(on):
    lib().go_set_strict_orthogonal(C.c_int(int(on)))


def set_accumulate_f64(on):
    """long sums in double instead of the reference's sequential f32 (see groan_oracle.h)"""
    lib().go_set_accumulate_f64(C.c_int(int(on)))


class acc64:
    """with O.acc64(): ... -> oracle sums in double inside the block"""

    def __enter__(self):
        set_accumulate_f64(True)

    def __exit__(self, *a):
        set_accumulate_f64(False)


# ---------------- primitives ----------------
def wrap(p, box):
    p = _f(p).copy(); b = _box(box)
    lib().go_wrap(_p(p), _p(b))
    return p


def vector_to(a, bpt, box):
    a = _f(a); bpt = _f(bpt); b = _box(box); out = np.zeros(3, np.float32)
    lib().go_vector_to(_p(a), _p(bpt), _p(b), _p(out))
    return out


def distance(a, bpt, dim, box):
    a = _f(a); bpt = _f(bpt); b = _box(box)
    return float(lib().go_distance(_p(a), _p(bpt), C.c_int(DIM[dim]), _p(b)))


def distance_naive(a, bpt, dim):
    a = _f(a); bpt = _f(bpt)
    return float(lib().go_distance_naive(_p(a), _p(bpt), C.c_int(DIM[dim])))


def box_center(box):
    b = _box(box); out = np.zeros(3, np.float32)
    lib().go_box_center(_p(b), _p(out))
    return out


def box_from_lengths_angles(lengths, angles):
    l = _f(lengths); a = _f(angles); out = np.zeros(9, np.float32)
    lib().go_box_from_lengths_angles(_p(l), _p(a), _p(out))
    return out


# ---------------- containers ----------------
def _blocks_out(n):
    return np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)


def container_from_indices(indices, n_atoms):
    idx = _u(indices); s, e = _blocks_out(idx.size)
    nb = lib().go_container_from_indices(_p(idx), C.c_size_t(idx.size), C.c_uint64(n_atoms), _p(s), _p(e))
    return np.stack([s[:nb], e[:nb]], 1)


def container_from_ranges(ranges, n_atoms):
    r = np.asarray(ranges, np.uint64).reshape(-1, 2)
    st, en = _u(r[:, 0]), _u(r[:, 1]); s, e = _blocks_out(r.shape[0])
    nb = lib().go_container_from_ranges(_p(st), _p(en), C.c_size_t(r.shape[0]), C.c_uint64(n_atoms), _p(s), _p(e))
    return np.stack([s[:nb], e[:nb]], 1)


def _se(blocks):
    b = np.asarray(blocks, np.uint64).reshape(-1, 2)
    return _u(b[:, 0]), _u(b[:, 1]), b.shape[0]


def container_union(b1, b2):
    s1, e1, n1 = _se(b1); s2, e2, n2 = _se(b2); s, e = _blocks_out(n1 + n2)
    nb = lib().go_container_union(_p(s1), _p(e1), C.c_size_t(n1), _p(s2), _p(e2), C.c_size_t(n2), _p(s), _p(e))
    return np.stack([s[:nb], e[:nb]], 1)


def container_intersection(b1, b2):
    s1, e1, n1 = _se(b1); s2, e2, n2 = _se(b2)
    cap = int(lib().go_container_n_atoms(_p(s1), _p(e1), C.c_size_t(n1))) + 1
    s, e = _blocks_out(cap)
    nb = lib().go_container_intersection(_p(s1), _p(e1), C.c_size_t(n1), _p(s2), _p(e2), C.c_size_t(n2), _p(s), _p(e))
    return np.stack([s[:nb], e[:nb]], 1)


def container_expand(blocks):
    s, e, n = _se(blocks)
    na = int(lib().go_container_n_atoms(_p(s), _p(e), C.c_size_t(n)))
    out = np.zeros(max(na, 1), np.uint64)
    m = lib().go_container_expand(_p(s), _p(e), C.c_size_t(n), _p(out))
    return out[:m]


def container_isin(blocks, index):
    s, e, n = _se(blocks)
    return bool(lib().go_container_isin(_p(s), _p(e), C.c_size_t(n), C.c_uint64(index)))


# ---------------- centres ----------------
def _center_call(fn, pos, mass, idx, box):
    pos = _f(pos); idx = _u(idx); out = np.zeros(3, np.float32); err = C.c_uint64(0)
    m = _f(mass) if mass is not None else None
    args = [_p(pos), C.c_size_t(12), _p(m), C.c_size_t(4), _p(idx), C.c_size_t(idx.size)]
    if box is not False:
        args.append(_p(_box(box)))
    st = fn(*args, _p(out), C.byref(err))
    if st != OK:
        raise OracleError(st, err.value)
    return out


def center_naive(pos, idx, mass=None):
    return _center_call(lib().go_center_naive, pos, mass, idx, False)


def estimate_center(pos, idx, box, mass=None):
    return _center_call(lib().go_estimate_center, pos, mass, idx, box)


def get_center(pos, idx, box, mass=None):
    return _center_call(lib().go_get_center, pos, mass, idx, box)


# ---------------- distances ----------------
def group_all_distances(pos, idx1, idx2, dim, box):
    pos = _f(pos); i1 = _u(idx1); i2 = _u(idx2); b = _box(box)
    out = np.zeros((i1.size, i2.size), np.float32); err = C.c_uint64(0)
    st = lib().go_group_all_distances(_p(pos), C.c_size_t(12), _p(i1), C.c_size_t(i1.size), _p(i2),
                                      C.c_size_t(i2.size), C.c_int(DIM[dim]), _p(b), _p(out), C.byref(err))
    if st != OK:
        raise OracleError(st, err.value)
    return out


# ---------------- translate / wrap / centre ----------------
def translate(pos, idx, v, box):
    pos = _f(pos).copy(); idx = _u(idx); v = _f(v); err = C.c_uint64(0)
    st = lib().go_translate(_p(pos), C.c_size_t(12), _p(idx), C.c_size_t(idx.size), _p(v), _p(_box(box)), C.byref(err))
    if st != OK:
        raise OracleError(st, err.value)
    return pos


def wrap_atoms(pos, idx, box):
    pos = _f(pos).copy(); idx = _u(idx); err = C.c_uint64(0)
    st = lib().go_wrap_atoms(_p(pos), C.c_size_t(12), _p(idx), C.c_size_t(idx.size), _p(_box(box)), C.byref(err))
    if st != OK:
        raise OracleError(st, err.value)
    return pos


def atoms_center(pos, ref_idx, dim, box, mass=None):
    pos = _f(pos).copy(); r = _u(ref_idx); a = np.arange(pos.shape[0], dtype=np.uint64); err = C.c_uint64(0)
    m = _f(mass) if mass is not None else None
    st = lib().go_atoms_center(_p(pos), C.c_size_t(12), _p(m), C.c_size_t(4), _p(r), C.c_size_t(r.size),
                               _p(a), C.c_size_t(a.size), C.c_int(DIM[dim]), C.c_int(int(mass is not None)),
                               _p(_box(box)), C.byref(err))
    if st != OK:
        raise OracleError(st, err.value)
    return pos


# ---------------- Kabsch / RMSD ----------------
def kabsch_rmsd(p, q, w, cp, cq, sum_w):
    p = _f(p); q = _f(q); w = _f(w); cp = _f(cp); cq = _f(cq)
    R = np.zeros(9, np.float32); t = np.zeros(3, np.float32); r = C.c_float(0)
    lib().go_kabsch_rmsd(_p(p), _p(q), _p(w), C.c_size_t(p.shape[0]), _p(cp), _p(cq), C.c_float(sum_w),
                         _p(R), _p(t), C.byref(r))
    # column-major storage -> R[i, j]
    return R.reshape(3, 3).T.copy(), t, r.value


def calc_rmsd(ref_pos, ref_mass, ref_idx, ref_box, cur_pos, cur_mass, cur_idx, cur_box):
    rp = _f(ref_pos); rm = _f(ref_mass); ri = _u(ref_idx); cp = _f(cur_pos); cm = _f(cur_mass); ci = _u(cur_idx)
    R = np.zeros(9, np.float32); r = C.c_float(0); err = C.c_uint64(0); cnt = (C.c_uint64 * 2)()
    st = lib().go_calc_rmsd(_p(rp), C.c_size_t(12), _p(rm), C.c_size_t(4), _p(ri), C.c_size_t(ri.size), _p(_box(ref_box)),
                            _p(cp), C.c_size_t(12), _p(cm), C.c_size_t(4), _p(ci), C.c_size_t(ci.size), _p(_box(cur_box)),
                            _p(R), C.byref(r), C.byref(err), cnt)
    if st != OK:
        raise OracleError(st, err.value, (cnt[0], cnt[1]))
    return r.value, R.reshape(3, 3).T.copy()


def calc_rmsd_and_fit(ref_pos, ref_mass, ref_idx, ref_box, cur_pos, cur_mass, cur_idx, cur_box):
    rp = _f(ref_pos); rm = _f(ref_mass); ri = _u(ref_idx)
    cp = _f(cur_pos).copy(); cm = _f(cur_mass); ci = _u(cur_idx)
    allidx = np.arange(cp.shape[0], dtype=np.uint64)
    r = C.c_float(0); err = C.c_uint64(0); cnt = (C.c_uint64 * 2)()
    st = lib().go_calc_rmsd_and_fit(_p(rp), C.c_size_t(12), _p(rm), C.c_size_t(4), _p(ri), C.c_size_t(ri.size), _p(_box(ref_box)),
                                    _p(cp), C.c_size_t(12), _p(cm), C.c_size_t(4), _p(ci), C.c_size_t(ci.size),
                                    _p(allidx), C.c_size_t(allidx.size), _p(_box(cur_box)),
                                    C.byref(r), C.byref(err), cnt)
    if st != OK:
        raise OracleError(st, err.value, (cnt[0], cnt[1]))
    return r.value, cp


def baseline_rmsd_fit(frames, ref_xyz, masses, box, n_threads, layout):
    """frames [F,N,3] fitted in place; returns (seconds, rmsd[F])."""
    assert frames.dtype == np.float32 and frames.flags["C_CONTIGUOUS"]
    F, N, _ = frames.shape
    ref = _f(ref_xyz); m = _f(masses); out = np.zeros(F, np.float32)
    sec = lib().go_baseline_rmsd_fit(_p(frames), C.c_size_t(F), C.c_size_t(N), _p(ref), _p(m), _p(_box(box)),
                                     C.c_int(n_threads), C.c_int(layout), _p(out))
    return float(sec), out


# ---------------- geometry selection ----------------
class GoShape(C.Structure):
    _fields_ = [("kind", C.c_int), ("position", C.c_float * 3), ("size", C.c_float * 3), ("base2", C.c_float * 3),
                ("base3", C.c_float * 3), ("orientation", C.c_int), ("plane", C.c_int)]


SHAPE_SPHERE, SHAPE_RECTANGULAR, SHAPE_CYLINDER, SHAPE_PRISM = 1, 2, 3, 4
_PLANE_OF = {"x": "yz", "y": "xz", "z": "xy"}


def shape(spec):
    """spec = dict(kind='sphere'|'rectangular'|'cylinder'|'prism', ...) as in tests/golden/shape_cases.json"""
    s = GoShape()
    k = spec["kind"]
    if k == "sphere":
        s.kind = SHAPE_SPHERE; s.position[:] = spec["position"]; s.size[0] = spec["radius"]
    elif k == "rectangular":
        s.kind = SHAPE_RECTANGULAR; s.position[:] = spec["position"]; s.size[:] = spec["size"]
    elif k == "cylinder":
        s.kind = SHAPE_CYLINDER; s.position[:] = spec["position"]; s.size[0] = spec["radius"]; s.size[1] = spec["height"]
        o = spec["orientation"].lower(); s.orientation = DIM[o]; s.plane = DIM[_PLANE_OF[o]]
    elif k == "prism":
        b1, b2, b3 = (_f(spec[n]) for n in ("base1", "base2", "base3"))
        st = lib().go_shape_prism_init(C.byref(s), _p(b1), _p(b2), _p(b3), C.c_float(spec["height"]))
        if st != 0:
            raise OracleError(100 + st)
    else:
        raise ValueError(k)
    return s


def shape_inside(spec, point, box):
    s = spec if isinstance(spec, GoShape) else shape(spec)
    return bool(lib().go_shape_inside(C.byref(s), _p(_f(point)), _p(_box(box))))


def shape_inside_naive(spec, point):
    s = spec if isinstance(spec, GoShape) else shape(spec)
    return lib().go_shape_inside_naive(C.byref(s), _p(_f(point))) == 1


def group_from_geometries(pos, idx, box, specs, naive=False):
    pos = _f(pos); idx = _u(idx)
    arr = (GoShape * len(specs))(*[s if isinstance(s, GoShape) else shape(s) for s in specs])
    out = np.zeros(idx.size, np.uint64)
    lib().go_group_from_geometries.restype = C.c_size_t
    n = lib().go_group_from_geometries(_p(pos), C.c_size_t(12), _p(idx), C.c_size_t(idx.size), _p(_box(box)), arr,
                                       C.c_size_t(len(specs)), C.c_int(int(naive)), _p(out))
    return out[:n]


# ---------------- cut-off pair search ----------------
def pairs_within(pos, idx1, idx2, box, cutoff):
    pos = _f(pos); i1 = _u(idx1); i2 = _u(idx2)
    L = lib(); L.go_pairs_within.restype = C.c_size_t
    args = lambda cap, oi, oj, od: (_p(pos), C.c_size_t(12), _p(i1), C.c_size_t(i1.size), _p(i2), C.c_size_t(i2.size), _p(_box(box)), C.c_float(cutoff),
                                    C.c_size_t(cap), _p(oi), _p(oj), _p(od))
    e = np.zeros(1, np.uint64); ef = np.zeros(1, np.float32)
    n = L.go_pairs_within(*args(0, e, e, ef))
    oi = np.zeros(max(n, 1), np.uint64); oj = np.zeros(max(n, 1), np.uint64); od = np.zeros(max(n, 1), np.float32)
    L.go_pairs_within(*args(n, oi, oj, od))
    return oi[:n], oj[:n], od[:n]
