"""Host-side mirror of the groan_rs `System` surface for the per-frame geometry path.

Same method names, argument meaning and error behaviour as the reference (src/system/{mod,analysis,
rmsd,modifying,utility}.rs); every method is a thin call into libgroan_hip.so through the C ABI
(include/groan_hip.h).  No arithmetic on atoms happens in Python.
"""
import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib
from ._lib import (OK, E_NO_BOX, E_NOT_ORTHOGONAL, E_ZERO_BOX, E_EMPTY_GROUP, E_INCONSISTENT_GROUP,
                   E_NO_POSITION, E_NO_MASS, E_GROUP_NOT_FOUND, E_OUT_OF_RANGE, E_GROUP_EXISTS, E_INVALID_NAME)


class Dimension(IntEnum):
    """src/structures/dimension.rs:13-23"""
    NONE = 0
    X = 1
    Y = 2
    Z = 3
    XY = 4
    XZ = 5
    YZ = 6
    XYZ = 7

    def is_x(self):
        return self in (Dimension.X, Dimension.XY, Dimension.XZ, Dimension.XYZ)

    def is_y(self):
        return self in (Dimension.Y, Dimension.XY, Dimension.YZ, Dimension.XYZ)

    def is_z(self):
        return self in (Dimension.Z, Dimension.XZ, Dimension.YZ, Dimension.XYZ)


# ----------------------------------------------------------------------------- errors (src/errors.rs)
class GroanError(Exception):
    """variant = name of the reference's enum variant; detail = its payload."""

    def __init__(self, variant, detail=None, status=None):
        super().__init__("%s::%s(%r)" % (type(self).__name__, variant, detail))
        self.variant, self.detail, self.status = variant, detail, status


class SimBoxError(GroanError):      # errors.rs:556-566
    pass


class GroupError(GroanError):       # errors.rs:227-256
    pass


class AtomError(GroanError):        # errors.rs:290-305
    pass


class RMSDError(GroanError):        # errors.rs:624-650
    pass


class DeviceError(GroanError):
    pass


def _simbox(status):
    return SimBoxError({E_NO_BOX: "DoesNotExist", E_NOT_ORTHOGONAL: "NotOrthogonal", E_ZERO_BOX: "ZeroLength"}.get(status, "Invalid"),
                       status=status)


def _as_box9(box):
    if box is None:
        return None
    b = np.ascontiguousarray(box, dtype=np.float32).ravel()
    if b.size == 3:
        b = np.concatenate([b, np.zeros(6, np.float32)])
    if b.size != 9:
        raise ValueError("box must have 3 or 9 elements (gro order)")
    if b[3] != 0.0 or b[4] != 0.0 or b[6] != 0.0:
        # SimBox::from panics on these (simbox.rs:37-39)
        raise ValueError("Unsupported Gromacs simulation box.")
    return np.ascontiguousarray(b)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ----------------------------------------------------------------------------- AtomContainer
class AtomContainer:
    """src/structures/container.rs -- sorted, merged, inclusive [start, end] blocks (bit-exact)."""

    def __init__(self, blocks=()):
        self.blocks = [(int(s), int(e)) for s, e in blocks]

    @staticmethod
    def _out(n):
        return np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64)

    @classmethod
    def from_indices(cls, indices, n_atoms):
        lib = _lib.load()
        idx = np.ascontiguousarray(list(indices), dtype=np.uint64)
        s, e = cls._out(idx.size)
        nb = lib.gr_container_from_indices(_ptr(idx), idx.size, n_atoms, _ptr(s), _ptr(e))
        return cls(zip(s[:nb], e[:nb]))

    @classmethod
    def from_ranges(cls, ranges, n_atoms):
        lib = _lib.load()
        r = np.asarray(list(ranges), dtype=np.uint64).reshape(-1, 2)
        st, en = np.ascontiguousarray(r[:, 0]), np.ascontiguousarray(r[:, 1])
        s, e = cls._out(r.shape[0])
        nb = lib.gr_container_from_ranges(_ptr(st), _ptr(en), r.shape[0], n_atoms, _ptr(s), _ptr(e))
        return cls(zip(s[:nb], e[:nb]))

    def _se(self):
        b = np.asarray(self.blocks, dtype=np.uint64).reshape(-1, 2)
        return np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1]), b.shape[0]

    @staticmethod
    def union(a, b):
        lib = _lib.load()
        s1, e1, n1 = a._se(); s2, e2, n2 = b._se()
        s, e = AtomContainer._out(n1 + n2)
        nb = lib.gr_container_union(_ptr(s1), _ptr(e1), n1, _ptr(s2), _ptr(e2), n2, _ptr(s), _ptr(e))
        return AtomContainer(zip(s[:nb], e[:nb]))

    @staticmethod
    def intersection(a, b):
        lib = _lib.load()
        s1, e1, n1 = a._se(); s2, e2, n2 = b._se()
        s, e = AtomContainer._out(a.get_n_atoms() + 1)
        nb = lib.gr_container_intersection(_ptr(s1), _ptr(e1), n1, _ptr(s2), _ptr(e2), n2, _ptr(s), _ptr(e))
        return AtomContainer(zip(s[:nb], e[:nb]))

    def get_n_atoms(self):
        s, e, n = self._se()
        return int(_lib.load().gr_container_n_atoms(_ptr(s), _ptr(e), n))

    def isin(self, index):
        s, e, n = self._se()
        return bool(_lib.load().gr_container_isin(_ptr(s), _ptr(e), n, index))

    def is_empty(self):
        return not self.blocks

    def first(self):
        return self.blocks[0][0] if self.blocks else None

    def last(self):
        return self.blocks[-1][1] if self.blocks else None

    def __iter__(self):
        s, e, n = self._se()
        out = np.zeros(max(self.get_n_atoms(), 1), np.uint64)
        m = _lib.load().gr_container_expand(_ptr(s), _ptr(e), n, _ptr(out))
        return iter(int(v) for v in out[:m])

    def __eq__(self, other):
        return isinstance(other, AtomContainer) and self.blocks == other.blocks

    def __repr__(self):
        return "AtomContainer(%r)" % (self.blocks,)


# ----------------------------------------------------------------------------- System
class AtomIterator:
    """Iterator-level surface of the reference: AtomIterator / MutAtomIterator / FilterAtomIterator / Union / Intersection
    (src/structures/iterators.rs:28-46,350-402,1053-1604) -- an AtomContainer + the system (and slot) whose box and atoms it
    walks.  Obtained from System.group_iter / atoms_iter / selection_iter / container_iter (src/system/iterating.rs:43-140).
    Every method is ONE call into the C ABI's anonymous-selection entry points (gr_sel_*): the kernels take the blocks directly.

    Error behaviour is the iterator traits', not System's: box first (AtomError::InvalidSimBox), then the first atom without
    position / mass; an EMPTY iterator is not an error, its centre is (NaN, NaN, NaN) (iterators.rs:1186-1188)."""

    def __init__(self, system, container, slot=0):
        self.system, self.container, self.slot = system, container, slot

    def _se(self):
        b = np.asarray(self.container.blocks, dtype=np.uint64).reshape(-1, 2)
        return np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1]), b.shape[0]

    def __iter__(self):
        return iter(self.container)

    def __len__(self):
        return self.container.get_n_atoms()

    def get_n_atoms(self):
        return self.container.get_n_atoms()

    def _center(self, kind, weighted):
        s, e, n = self._se()
        out = np.zeros(3, np.float32)
        st = self.system._lib.gr_sel_center(self.system._ctx, self.slot, _ptr(s), _ptr(e), n, kind, weighted, _ptr(out))
        if st != OK:
            self.system._raise_atom(st)
        return out

    def get_center_naive(self): return self._center(_lib.CENTER_NAIVE, 0)       # iterators.rs:886-903
    def get_com_naive(self): return self._center(_lib.CENTER_NAIVE, 1)          # :946-967
    def estimate_center(self): return self._center(_lib.CENTER_ESTIMATE, 0)     # :1152-1191
    def get_center(self): return self._center(_lib.CENTER_PBC, 0)               # :1237-1266
    def estimate_com(self): return self._center(_lib.CENTER_ESTIMATE, 1)        # :1314-1357
    def get_com(self): return self._center(_lib.CENTER_PBC, 1)                  # :1404-1438

    def _filter(self, geometries, naive):
        from .shapes import pack
        arr, ns = pack(geometries if isinstance(geometries, (list, tuple)) else [geometries])
        s, e, n = self._se()
        nb, na = C.c_size_t(0), C.c_uint64(0)
        cap = max(self.get_n_atoms(), 1)
        os_, oe = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        st = self.system._lib.gr_sel_filter_geometry(self.system._ctx, self.slot, _ptr(s), _ptr(e), n, arr, ns, int(naive), _ptr(os_), _ptr(oe), cap,
                                                     C.byref(nb), C.byref(na))
        if st != OK:
            self.system._raise_atom(st)
        return AtomIterator(self.system, AtomContainer(list(zip(os_[:nb.value].tolist(), oe[:nb.value].tolist()))), self.slot)

    def filter_geometry(self, geometry):
        """AtomIteratorWithBox::filter_geometry (:1094-1105): atoms inside the shape, PBC-aware; the reference panics without a box"""
        return self._filter(geometry, False)

    def filter_geometry_naive(self, geometry):
        """ImmutableAtomIterable::filter_geometry_naive (:994-1004)"""
        return self._filter(geometry, True)

    def translate(self, vector):
        """MutAtomIteratorWithBox::translate (:1520-1524)"""
        s, e, n = self._se()
        v = np.ascontiguousarray(vector, dtype=np.float32)
        st = self.system._lib.gr_sel_translate(self.system._ctx, self.slot, _ptr(s), _ptr(e), n, _ptr(v))
        if st != OK:
            self.system._raise_atom(st)

    def wrap(self):
        """MutAtomIteratorWithBox::wrap (:1548-1553)"""
        s, e, n = self._se()
        st = self.system._lib.gr_sel_wrap(self.system._ctx, self.slot, _ptr(s), _ptr(e), n)
        if st != OK:
            self.system._raise_atom(st)

    def union(self, other):
        """OrderedAtomIterator::union (:1572-1591): every atom once, in index order"""
        return AtomIterator(self.system, AtomContainer.union(self.container, other.container), self.slot)

    def intersection(self, other):
        """OrderedAtomIterator::intersection (:1593-1604)"""
        return AtomIterator(self.system, AtomContainer.intersection(self.container, other.container), self.slot)

    def all_distances(self, other, dim=None):
        """the double loop of System::group_all_distances (analysis.rs:414-424) over two iterators -> [n1, n2] float32"""
        s1, e1, n1 = self._se(); s2, e2, n2 = other._se()
        out = np.zeros((self.get_n_atoms(), other.get_n_atoms()), np.float32)
        st = self.system._lib.gr_sel_all_distances(self.system._ctx, self.slot, _ptr(s1), _ptr(e1), n1, _ptr(s2), _ptr(e2), n2,
                                                   int(Dimension.XYZ if dim is None else dim), _ptr(out), out.size)
        if st != OK:
            self.system._raise_atom(st)
        return out


class System:
    """Device mirror of groan_rs `System` (src/system/mod.rs:38-73): atoms' masses, named groups,
    `n_slots` resident frames (positions + box).  Slot 0 is "the current frame"."""

    def __init__(self, n_atoms, masses=None, box=None, positions=None, device=0, n_slots=1, name="System"):
        self._lib = _lib.load()
        st = C.c_int(0)
        self._ctx = self._lib.gr_ctx_create(int(device), int(n_atoms), int(n_slots), C.byref(st))
        if not self._ctx:
            raise DeviceError("ContextCreation", self._lib.gr_status_string(st.value).decode(), st.value)
        self.name, self.n_atoms, self.n_slots, self.device = name, int(n_atoms), int(n_slots), int(device)
        self.simulation_step, self.simulation_time = 0, 0.0
        self._plans = []
        if masses is not None:
            self.set_masses(masses)
        if positions is not None:
            self.set_frame(positions, box)
        elif box is not None:
            self.set_box(box)

    @classmethod
    def _borrow(cls, ctx, n_atoms, n_slots, device, name="System"):
        """a System over a context somebody else owns (the workers of a gr_pool): close() releases the plans, not the context"""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._ctx, self._owned = ctx, False
        self.name, self.n_atoms, self.n_slots, self.device = name, int(n_atoms), int(n_slots), int(device)
        self.simulation_step, self.simulation_time = 0, 0.0
        self._plans = []
        return self

    # -- lifetime
    def close(self):
        if getattr(self, "_ctx", None):
            for p in self._plans:
                p.close()
            if getattr(self, "_owned", True):
                self._lib.gr_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- error mapping
    def _err(self, status):
        lib = self._lib
        msg = lib.gr_last_error(self._ctx).decode(errors="replace")
        return status, msg, int(lib.gr_last_error_index(self._ctx))

    def _raise_group(self, status):
        st, msg, idx = self._err(status)
        if st == E_GROUP_NOT_FOUND: raise GroupError("NotFound", msg, st)
        if st == E_INVALID_NAME: raise GroupError("InvalidName", msg, st)
        if st == E_EMPTY_GROUP: raise GroupError("EmptyGroup", msg, st)
        if st in (E_NO_BOX, E_NOT_ORTHOGONAL, E_ZERO_BOX): raise GroupError("InvalidSimBox", _simbox(st), st)
        if st == E_NO_POSITION: raise GroupError("InvalidPosition", idx, st)
        if st == E_NO_MASS: raise GroupError("InvalidMass", idx, st)
        if st == E_OUT_OF_RANGE: raise AtomError("OutOfRange", idx, st)   # a selection outside the system (the reference panics on first use)
        raise DeviceError(self._lib.gr_status_string(st).decode(), msg, st)

    def _raise_atom(self, status):
        st, msg, idx = self._err(status)
        if st in (E_NO_BOX, E_NOT_ORTHOGONAL, E_ZERO_BOX): raise AtomError("InvalidSimBox", _simbox(st), st)
        if st == E_NO_POSITION: raise AtomError("InvalidPosition", idx, st)
        if st == E_NO_MASS: raise AtomError("InvalidMass", idx, st)
        if st == E_OUT_OF_RANGE: raise AtomError("OutOfRange", idx, st)
        raise DeviceError(self._lib.gr_status_string(st).decode(), msg, st)

    def _raise_rmsd(self, status):
        st, msg, idx = self._err(status)
        if st == E_GROUP_NOT_FOUND: raise RMSDError("NonexistentGroup", msg, st)
        if st == E_EMPTY_GROUP: raise RMSDError("EmptyGroup", msg, st)
        if st in (E_NO_BOX, E_NOT_ORTHOGONAL, E_ZERO_BOX): raise RMSDError("InvalidSimBox", _simbox(st), st)
        if st == E_NO_POSITION: raise RMSDError("InvalidPosition", idx, st)
        if st == E_NO_MASS: raise RMSDError("InvalidMass", idx, st)
        if st == E_INCONSISTENT_GROUP:
            cnt = (C.c_uint64 * 2)()
            self._lib.gr_last_error_counts(self._ctx, cnt)
            raise RMSDError("InconsistentGroup", (msg, int(cnt[0]), int(cnt[1])), st)
        raise DeviceError(self._lib.gr_status_string(st).decode(), msg, st)

    # -- static data
    def get_n_atoms(self):
        return self.n_atoms

    def set_masses(self, masses):
        m = np.ascontiguousarray(masses, dtype=np.float32)
        st = self._lib.gr_set_masses(self._ctx, _ptr(m), m.size)
        if st != OK:
            raise DeviceError("set_masses", self._err(st)[1], st)

    TUNE = {"sub_batch": 1, "chunks": 2, "fit_wgs": 3, "fuse": 4, "two_pass": 5, "resident": 6, "resident_groups": 7, "resident_streams": 8, "resident_fill": 9, "pairdist_symmetric": 10, "resident_wg_groups": 11, "rmsd_fast": 12, "rmsd_fast_min": 13, "rmsd_fast_sigmas": 14, "masked_selections": 15, "xtc_device_encode": 16, "small_calls": 17, "resident_metro_ns": 18, "resident_fit_last": 19, "stream_wgs_per_cu": 20, "center_resident": 21, "translate_rows": 22, "test_resident_no_start": 100, "test_resident_abort_at": 101}

    def set_tuning(self, **kw):
        """gr_ctx_set_tuning: launch geometry / path selection of the batched RMSD calls (measurement only; same results)"""
        for k, v in kw.items():
            st = self._lib.gr_ctx_set_tuning(self._ctx, self.TUNE[k], int(v))
            if st != OK:
                raise DeviceError("set_tuning", self._err(st)[1], st)

    STAT = {"n_cus": 1, "res_max_wgs": 2, "res_launches": 3, "res_handshake_misses": 4, "res_aborts": 5, "res_redone_frames": 6, "res_last_streams": 7, "rmsd_fast_frames": 8, "rmsd_exact_redos": 9, "xtc_device_frames": 10, "small_calls": 11, "small_sync_fallbacks": 12, "res_metro_period_ns": 13, "res_last_turn_ns": 14, "res_late_permille": 15, "res_sclk_mhz": 16, "center_res_launches": 17, "center_res_redone": 18}

    def stat(self, key):
        """gr_ctx_stat: device facts and counters of the batched RMSD path"""
        v = C.c_uint64(0)
        st = self._lib.gr_ctx_stat(self._ctx, self.STAT[key], C.byref(v))
        if st != OK:
            raise DeviceError("stat", self._err(st)[1], st)
        return int(v.value)

    def set_center_onepass_min(self, min_atoms):
        """get_center / get_com of contiguous groups of at least `min_atoms` atoms in one pass (0 = always two passes)"""
        self._lib.gr_ctx_set_center_onepass_min(self._ctx, int(min_atoms))

    def center_fallbacks(self):
        """frames the one-pass centre handed to the two-pass path so far"""
        return int(self._lib.gr_center_fallbacks(self._ctx))

    def set_strict_orthogonal(self, on=True):
        """reproduce the reference's SimBoxError::NotOrthogonal for non-orthogonal boxes"""
        self._strict_flag = bool(on)
        self._lib.gr_ctx_set_strict_orthogonal(self._ctx, int(bool(on)))

    # -- groups (src/system/groups.rs)
    def group_create_from_ranges(self, name, ranges):
        r = np.asarray(list(ranges), dtype=np.uint64).reshape(-1, 2)
        s, e = np.ascontiguousarray(r[:, 0]), np.ascontiguousarray(r[:, 1])
        st = self._lib.gr_group_create_from_ranges(self._ctx, name.encode(), _ptr(s), _ptr(e), r.shape[0])
        if st not in (OK, E_GROUP_EXISTS):
            self._raise_group(st)
        return st == E_GROUP_EXISTS   # the reference returns AlreadyExistsWarning

    def group_create_from_indices(self, name, indices):
        idx = np.ascontiguousarray(list(indices), dtype=np.uint64)
        st = self._lib.gr_group_create_from_indices(self._ctx, name.encode(), _ptr(idx), idx.size)
        if st not in (OK, E_GROUP_EXISTS):
            self._raise_group(st)
        return st == E_GROUP_EXISTS

    def group_create_from_geometry(self, name, source_group, geometry, slot=0, naive=False):
        """System::group_create_from_geometry (groups.rs:94-118) with a source GROUP in place of the query string"""
        return self.group_create_from_geometries(name, source_group, [geometry], slot=slot, naive=naive)

    def group_create_from_geometries(self, name, source_group, geometries, slot=0, naive=False):
        """System::group_create_from_geometries (groups.rs:164-188): atoms of `source_group` that have a position and
        lie inside every shape, in the source order; needs an orthogonal box"""
        from .shapes import pack
        arr, n = pack(geometries)
        st = self._lib.gr_group_create_from_geometries(self._ctx, slot, name.encode(), source_group.encode(), arr, n, int(bool(naive)))
        if st == E_GROUP_NOT_FOUND:
            raise GroupError("InvalidQuery", ("GroupNotFound", source_group), st)      # SelectError::GroupNotFound
        if st not in (OK, E_GROUP_EXISTS):
            self._raise_group(st)
        return st == E_GROUP_EXISTS

    def group_create_from_container(self, name, container):
        return self.group_create_from_ranges(name, container.blocks)

    def group_remove(self, name):
        st = self._lib.gr_group_remove(self._ctx, name.encode())
        if st != OK:
            self._raise_group(st)

    def group_exists(self, name):
        return bool(self._lib.gr_group_exists(self._ctx, name.encode()))

    def group_names(self):
        out, buf = [], C.create_string_buffer(256)
        for i in range(int(self._lib.gr_group_count(self._ctx))):
            if self._lib.gr_group_name(self._ctx, i, buf, 256) == OK:
                out.append(buf.value.decode(errors="replace"))
        return out

    def group_create(self, name, query, structure):
        """System::group_create (groups.rs:36-92): a group from a selection-language query (groan_rs_amd/select.py) over the
        names / numbers of `structure` (textio.Structure); -> True when an existing group was overwritten"""
        from .select import group_create as _group_create   # (the package re-exports the FUNCTION `select` under the module's name)
        return _group_create(self, name, query, structure)

    def group_get_n_atoms(self, name):
        n = C.c_uint64(0)
        st = self._lib.gr_group_n_atoms(self._ctx, name.encode(), C.byref(n))
        if st != OK:
            raise GroupError("NotFound", name, st)
        return int(n.value)

    def group_isempty(self, name):
        return self.group_get_n_atoms(name) == 0

    # -- iterators (src/system/iterating.rs:43-140)
    def group_iter(self, name, slot=0):
        """System::group_iter / group_iter_mut: GroupError::NotFound for an unknown group"""
        return AtomIterator(self, self.group_container(name), slot)

    def atoms_iter(self, slot=0):
        """System::atoms_iter / atoms_iter_mut"""
        return AtomIterator(self, AtomContainer([(0, self.n_atoms - 1)]), slot)

    def container_iter(self, container, slot=0):
        return AtomIterator(self, container, slot)

    def selection_iter(self, query, structure, slot=0):
        """System::selection_iter / selection_iter_mut (iterating.rs:124-140): the atoms a selection-language query picks"""
        import importlib
        sel = importlib.import_module(".select", __package__)    # (the package re-exports the FUNCTION `select` under the module's name)
        groups = {g: np.array(list(self.group_container(g)), np.int64) for g in self.group_names()}
        idx = sel.select(structure, query, groups)
        return AtomIterator(self, AtomContainer.from_indices([int(i) for i in idx], self.n_atoms), slot)

    def group_container(self, name):
        nb = C.c_size_t(0)
        st = self._lib.gr_group_n_blocks(self._ctx, name.encode(), C.byref(nb))
        if st != OK:
            raise GroupError("NotFound", name, st)
        s, e = np.zeros(max(nb.value, 1), np.uint64), np.zeros(max(nb.value, 1), np.uint64)
        self._lib.gr_group_blocks(self._ctx, name.encode(), _ptr(s), _ptr(e))
        return AtomContainer(zip(s[:nb.value], e[:nb.value]))

    # -- frames
    def set_frame(self, positions, box="keep", slot=0, step=None, time=None):
        """TrajRead::update_system: positions float32 [n_atoms,3] as delivered by the xtc readers."""
        x = np.ascontiguousarray(positions, dtype=np.float32)
        if x.shape != (self.n_atoms, 3):
            raise ValueError("positions must have shape (n_atoms, 3)")
        if isinstance(box, str) and box == "keep":
            b = self.get_box(slot)
        else:
            b = _as_box9(box)
        st = self._lib.gr_frame_upload(self._ctx, slot, _ptr(x), _ptr(b))
        if st != OK:
            raise DeviceError("frame_upload", self._err(st)[1], st)
        self._lib.gr_sync(self._ctx)   # x may be a temporary
        if step is not None: self.simulation_step = step
        if time is not None: self.simulation_time = time

    def upload_async(self, host_array, box, slot):
        """gr_frame_upload without waiting: host_array must stay valid (use pinned_array) until upload_wait/sync"""
        st = self._lib.gr_frame_upload(self._ctx, slot, _ptr(host_array), _ptr(_as_box9(box)))
        if st != OK:
            raise DeviceError("frame_upload", self._err(st)[1], st)

    def upload_wait(self, slot):
        self._lib.gr_frame_upload_wait(self._ctx, slot)

    def get_positions(self, slot=0):
        out = np.empty((self.n_atoms, 3), np.float32)
        st = self._lib.gr_frame_download(self._ctx, slot, _ptr(out))
        if st != OK:
            raise DeviceError("frame_download", self._err(st)[1], st)
        return out

    def set_box(self, box, slot=0):
        st = self._lib.gr_frame_set_box(self._ctx, slot, _ptr(_as_box9(box)))
        if st != OK:
            raise DeviceError("set_box", self._err(st)[1], st)

    def reset_box(self, slot=0):
        self.set_box(None, slot)

    def get_box(self, slot=0):
        b = np.zeros(9, np.float32)
        st = self._lib.gr_frame_get_box(self._ctx, slot, _ptr(b))
        return b if st == OK else None

    def has_box(self, slot=0):
        return self.get_box(slot) is not None

    def get_box_center(self, slot=0):
        """src/system/mod.rs:298-308"""
        b = self.get_box(slot)
        if b is None:
            raise _simbox(E_NO_BOX)
        ortho = b[5] == 0.0 and b[7] == 0.0 and b[8] == 0.0
        if not ortho and self._strict():
            raise _simbox(E_NOT_ORTHOGONAL)
        two = np.float32(2.0)
        return np.array([b[0] / two, b[1] / two, b[2] / two], np.float32)

    def _strict(self):
        return getattr(self, "_strict_flag", False)

    def copy_frame(self, dst_slot, src_slot):
        st = self._lib.gr_frame_copy(self._ctx, dst_slot, src_slot)
        if st != OK:
            raise DeviceError("frame_copy", self._err(st)[1], st)

    # -- centres (src/system/analysis.rs:52-320)
    def _center(self, name, kind, weighted, slot):
        out = np.zeros(3, np.float32)
        st = self._lib.gr_group_center(self._ctx, slot, name.encode(), kind, int(weighted), _ptr(out))
        if st != OK:
            self._raise_group(st)
        return out

    def group_get_center_naive(self, name, slot=0): return self._center(name, _lib.CENTER_NAIVE, 0, slot)
    def group_estimate_center(self, name, slot=0): return self._center(name, _lib.CENTER_ESTIMATE, 0, slot)
    def group_get_center(self, name, slot=0): return self._center(name, _lib.CENTER_PBC, 0, slot)
    def group_get_com_naive(self, name, slot=0): return self._center(name, _lib.CENTER_NAIVE, 1, slot)
    def group_estimate_com(self, name, slot=0): return self._center(name, _lib.CENTER_ESTIMATE, 1, slot)
    def group_get_com(self, name, slot=0): return self._center(name, _lib.CENTER_PBC, 1, slot)

    def group_pairs_within(self, group1, group2, cutoff, slot=0):
        """all pairs (i in group1, j in group2, i != j) with distance <= cutoff through a device cell grid
        (CellGrid + distance filter, cellgrid.rs:301-409 / hbonds.rs:248-265) -> (i uint32[n], j uint32[n], d float32[n]) by i then j"""
        n = C.c_uint64(0)
        st = self._lib.gr_group_pairs_within(self._ctx, slot, group1.encode(), group2.encode(), C.c_float(cutoff), 0, None, None, None, C.byref(n))
        if st != OK:
            self._raise_group(st)
        m = int(n.value)
        i = np.zeros(max(m, 1), np.uint32); j = np.zeros(max(m, 1), np.uint32); d = np.zeros(max(m, 1), np.float32)
        if m:
            st = self._lib.gr_group_pairs_within(self._ctx, slot, group1.encode(), group2.encode(), C.c_float(cutoff), m, _ptr(i), _ptr(j), _ptr(d), C.byref(n))
            if st != OK:
                self._raise_group(st)
        return i[:m], j[:m], d[:m]

    # -- the same over a batch of resident frames (gr_*_batch): one set of launches, one read-back
    def group_center_batch(self, name, kind, weighted, first_slot, n_frames, raise_on_error=True):
        """-> (centres float32 [n_frames, 3] (NaN rows for failed frames), status int32 [n_frames])"""
        out = np.zeros((n_frames, 3), np.float32); st_arr = np.zeros(n_frames, np.int32)
        st = self._lib.gr_group_center_batch(self._ctx, first_slot, n_frames, name.encode(), kind, int(bool(weighted)), _ptr(out), _ptr(st_arr))
        if st != OK and raise_on_error:
            self._raise_group(st)
        return out, st_arr

    def group_get_com_batch(self, name, first_slot, n_frames, **kw): return self.group_center_batch(name, _lib.CENTER_PBC, 1, first_slot, n_frames, **kw)
    def group_get_center_batch(self, name, first_slot, n_frames, **kw): return self.group_center_batch(name, _lib.CENTER_PBC, 0, first_slot, n_frames, **kw)
    def group_estimate_com_batch(self, name, first_slot, n_frames, **kw): return self.group_center_batch(name, _lib.CENTER_ESTIMATE, 1, first_slot, n_frames, **kw)

    def atoms_center_batch(self, reference, first_slot, n_frames, dimension=Dimension.XYZ, weighted=False, raise_on_error=True):
        st_arr = np.zeros(n_frames, np.int32)
        st = self._lib.gr_atoms_center_batch(self._ctx, first_slot, n_frames, reference.encode(), int(dimension), int(bool(weighted)), _ptr(st_arr))
        if st != OK and raise_on_error:
            self._raise_group(st)
        return st_arr

    def group_wrap_batch(self, name, first_slot, n_frames, raise_on_error=True):
        st_arr = np.zeros(n_frames, np.int32)
        st = self._lib.gr_group_wrap_batch(self._ctx, first_slot, n_frames, name.encode() if name else None, _ptr(st_arr))
        if st != OK and raise_on_error:
            self._raise_group(st)
        return st_arr

    def group_translate_batch(self, name, vector, first_slot, n_frames, raise_on_error=True):
        v = np.ascontiguousarray(vector, np.float32); st_arr = np.zeros(n_frames, np.int32)
        st = self._lib.gr_group_translate_batch(self._ctx, first_slot, n_frames, name.encode() if name else None, _ptr(v), _ptr(st_arr))
        if st != OK and raise_on_error:
            self._raise_group(st)
        return st_arr

    # -- distances (analysis.rs:348-471)
    def group_distance(self, group1, group2, dim=Dimension.XYZ, slot=0):
        out = C.c_float(0)
        st = self._lib.gr_group_distance(self._ctx, slot, group1.encode(), group2.encode(), int(dim), C.byref(out))
        if st != OK:
            self._raise_group(st)
        return out.value

    def group_all_distances(self, group1, group2, dim=Dimension.XYZ, slot=0):
        n1, n2 = self.group_get_n_atoms(group1), self.group_get_n_atoms(group2)
        out = np.zeros((n1, n2), np.float32)
        st = self._lib.gr_group_all_distances(self._ctx, slot, group1.encode(), group2.encode(), int(dim), _ptr(out), out.size)
        if st != OK:
            self._raise_group(st)
        return out

    def group_all_distances_batch_device(self, group1, group2, first_slot, n_frames, dim=Dimension.XYZ, raise_on_error=True):
        """group_all_distances for `n_frames` consecutive resident slots in one launch; the matrices stay in HBM.
        -> (device pointer, n1, n2, status[n_frames]); matrix f starts f * n1 * n2 floats in; `device_read` fetches"""
        dev = C.c_void_p(); n1 = C.c_uint64(); n2 = C.c_uint64()
        status = np.zeros(n_frames, np.int32)
        st = self._lib.gr_group_all_distances_batch_device(self._ctx, first_slot, n_frames, group1.encode(), group2.encode(), int(dim),
                                                           C.byref(dev), C.byref(n1), C.byref(n2), _ptr(status))
        if st != OK and raise_on_error:
            self._raise_group(st)
        return dev, int(n1.value), int(n2.value), status

    PD_MIN, PD_MAX, PD_COUNT_BELOW, PD_HIST = 1, 2, 3, 4

    def group_all_distances_reduce(self, group1, group2, op, dim=Dimension.XYZ, per_row=False, param=0.0, nbins=0, first_slot=0, n_frames=1, raise_on_error=True):
        """what the callers of group_all_distances do with its matrix (analysis.rs:401-427; :1420-1451), without the matrix: `op` = "min" | "max"
        (float32), "count_below" (entries < param, uint64), "hist" (nbins bins over [0, param), uint64); per_row: one value per atom of
        group1 instead of one per frame.  -> (array [n_frames, len], status[n_frames]); equal to the reduction of the full matrices"""
        opc = {"min": self.PD_MIN, "max": self.PD_MAX, "count_below": self.PD_COUNT_BELOW, "hist": self.PD_HIST}[op]
        length = nbins if opc == self.PD_HIST else (self.group_get_n_atoms(group1) if per_row else 1)
        out = np.zeros((n_frames, max(length, 1)), np.float32 if opc in (self.PD_MIN, self.PD_MAX) else np.uint64)
        status = np.zeros(n_frames, np.int32)
        st = self._lib.gr_group_all_distances_reduce_batch(self._ctx, first_slot, n_frames, group1.encode(), group2.encode(), int(dim), opc, int(bool(per_row)),
                                                           C.c_float(param), int(nbins), _ptr(out), out.nbytes, _ptr(status))
        if st != OK and raise_on_error:
            self._raise_group(st)
        return out, status

    def device_read(self, dev, offset_floats, shape):
        """host copy of `shape` float32 values starting `offset_floats` into a device buffer handed out by the library"""
        out = np.zeros(shape, np.float32)
        st = self._lib.gr_device_read(self._ctx, C.c_void_p(dev.value + 4 * int(offset_floats)), _ptr(out), out.nbytes)
        if st != OK:
            self._raise_group(st)
        return out

    def atoms_distance(self, index1, index2, dim=Dimension.XYZ, slot=0):
        out = C.c_float(0)
        st = self._lib.gr_atoms_distance(self._ctx, slot, index1, index2, int(dim), C.byref(out))
        if st != OK:
            self._raise_atom(st)
        return out.value

    # -- translate / wrap / centre (modifying.rs:45-75,201-222; utility.rs:109-185)
    def atoms_translate(self, vector, slot=0):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        st = self._lib.gr_group_translate(self._ctx, slot, None, _ptr(v))
        if st != OK:
            self._raise_atom(st)

    def group_translate(self, name, vector, slot=0):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        st = self._lib.gr_group_translate(self._ctx, slot, name.encode(), _ptr(v))
        if st != OK:
            self._raise_group(st)

    def atoms_wrap(self, slot=0):
        st = self._lib.gr_group_wrap(self._ctx, slot, None)
        if st != OK:
            self._raise_atom(st)

    def group_wrap(self, name, slot=0):
        st = self._lib.gr_group_wrap(self._ctx, slot, name.encode())
        if st != OK:
            self._raise_group(st)

    def atoms_center(self, reference, dimension=Dimension.XYZ, slot=0):
        st = self._lib.gr_atoms_center(self._ctx, slot, reference.encode(), int(dimension), 0)
        if st != OK:
            self._raise_group(st)

    def atoms_center_mass(self, reference, dimension=Dimension.XYZ, slot=0):
        st = self._lib.gr_atoms_center(self._ctx, slot, reference.encode(), int(dimension), 1)
        if st != OK:
            self._raise_group(st)

    # -- RMSD (src/system/rmsd.rs:75-166)
    def calc_rmsd(self, reference, group, slot=0, ref_slot=0, return_rotation=False):
        r = C.c_float(0)
        R = np.zeros(9, np.float32)
        st = self._lib.gr_calc_rmsd(self._ctx, slot, reference._ctx, ref_slot, group.encode(), C.byref(r), _ptr(R))
        if st != OK:
            self._raise_rmsd(st)
        return (r.value, R.reshape(3, 3).T.copy()) if return_rotation else r.value

    def calc_rmsd_and_fit(self, reference, group, slot=0, ref_slot=0):
        r = C.c_float(0)
        st = self._lib.gr_calc_rmsd_and_fit(self._ctx, slot, reference._ctx, ref_slot, group.encode(), C.byref(r))
        if st != OK:
            self._raise_rmsd(st)
        return r.value

    # -- measurement helpers
    def sync(self):
        self._lib.gr_sync(self._ctx)

    def timer_start(self):
        self._lib.gr_timer_start(self._ctx)

    def timer_stop(self):
        ms = C.c_float(0)
        self._lib.gr_timer_stop(self._ctx, C.byref(ms))
        return ms.value

    def synth_reference(self, slot, box, radius, seed):
        st = self._lib.gr_synth_reference(self._ctx, slot, _ptr(_as_box9(box)), C.c_float(radius), seed)
        if st != OK:
            raise DeviceError("synth_reference", self._err(st)[1], st)

    def profile_enable(self, on=True):
        self._lib.gr_profile_enable(self._ctx, int(bool(on)))

    def profile_read(self):
        """-> {kernel: (ms_total, launches, frames)} for the batched RMSD path"""
        out = {}
        for k, name in enumerate(("k_sums_pk", "k_rmsd_finalize", "k_fit_pk", "k_fit_resident")):
            ms = C.c_double(0); n = C.c_uint64(0); f = C.c_uint64(0)
            self._lib.gr_profile_read(self._ctx, k, C.byref(ms), C.byref(n), C.byref(f))
            out[name] = (ms.value, int(n.value), int(f.value))
        return out

    def synth_frames(self, ref_slot, first_slot, n_frames, first_frame_index, noise_sigma, seed, frame_index_stride=1):
        st = self._lib.gr_synth_frames(self._ctx, ref_slot, first_slot, n_frames, first_frame_index, frame_index_stride,
                                       C.c_float(noise_sigma), seed)
        if st != OK:
            raise DeviceError("synth_frames", self._err(st)[1], st)

    def synth_uniform(self, slot, box, seed):
        st = self._lib.gr_synth_uniform(self._ctx, slot, _ptr(_as_box9(box)), seed)
        if st != OK:
            raise DeviceError("synth_uniform", self._err(st)[1], st)


def pinned_array(shape, dtype=np.float32):
    """numpy array over pinned host memory (gr_host_alloc): asynchronous H2D source for upload_async"""
    lib = _lib.load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib.gr_host_alloc(n)
    if not ptr:
        raise MemoryError("gr_host_alloc failed")
    buf = (C.c_char * n).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    return arr, ptr


def pinned_free(ptr):
    _lib.load().gr_host_free(ptr)


class RMSDPlan:
    """Cached reference side of the RMSD calculation = RMSDConverterAnalyzer::new (rmsd.rs:186-203)."""

    def __init__(self, reference, target, group, ref_slot=0):
        self._lib = _lib.load()
        st = C.c_int(0)
        self.target, self.group = target, group
        self._plan = self._lib.gr_rmsd_plan_create(reference._ctx, ref_slot, target._ctx, group.encode(), C.byref(st))
        if not self._plan:
            reference._raise_rmsd(st.value)
        target._plans.append(self)

    def close(self):
        if getattr(self, "_plan", None):
            self._lib.gr_rmsd_plan_destroy(self._plan)
            self._plan = None

    def force_exact(self, on=True):
        self._lib.gr_rmsd_plan_force_exact(self._plan, int(bool(on)))

    def last_fallbacks(self):
        return int(self._lib.gr_rmsd_plan_last_fallbacks(self._plan))

    def rmsd(self, first_slot=0, n_frames=1, return_rotation=False, raise_on_error=True):
        r = np.zeros(n_frames, np.float32); s = np.zeros(n_frames, np.int32); R = np.zeros((n_frames, 9), np.float32)
        st = self._lib.gr_rmsd_batch(self._plan, first_slot, n_frames, _ptr(r), _ptr(s), _ptr(R))
        if st != OK and raise_on_error:
            self.target._raise_rmsd(st)
        if return_rotation:
            return r, s, R.reshape(n_frames, 3, 3).transpose(0, 2, 1).copy()
        return r, s

    def begin(self, first_slot, n_frames, fit):
        """issue the whole batch and return at once (gr_rmsd_batch_begin); pair with end()"""
        st = self._lib.gr_rmsd_batch_begin(self._plan, first_slot, n_frames, int(bool(fit)))
        if st != OK:
            self.target._raise_rmsd(st)
        self._pending_n = n_frames

    def end(self, raise_on_error=True):
        n = self._pending_n
        r = np.zeros(n, np.float32); s = np.zeros(n, np.int32)
        st = self._lib.gr_rmsd_batch_end(self._plan, _ptr(r), _ptr(s), None)
        if st != OK and raise_on_error:
            self.target._raise_rmsd(st)
        return r, s

    def rmsd_fit(self, first_slot=0, n_frames=1, raise_on_error=True):
        r = np.zeros(n_frames, np.float32); s = np.zeros(n_frames, np.int32)
        st = self._lib.gr_rmsd_fit_batch(self._plan, first_slot, n_frames, _ptr(r), _ptr(s))
        if st != OK and raise_on_error:
            self.target._raise_rmsd(st)
        return r, s
