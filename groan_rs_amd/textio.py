"""Text front end: gro structures and ndx index groups over gr_gro_read / gr_ndx_* (groan_rs_amd/csrc/gr_textio.h) -- the
mirror of read_gro / System::from_file for .gro (src/io/gro_io/structure.rs:120-231) and System::read_ndx
(src/io/ndx_io.rs:31-230).  Errors carry the reference's variant names and payloads."""
import ctypes as C

import numpy as np

from . import _lib

_GRO_VARIANT = {1: "FileNotFound", 2: "LineNotFound", 3: "ParseLineErr", 4: "ParseAtomLineErr", 5: "ParseBoxLineErr", 6: "UnsupportedBox", 7: "InvalidFloat"}
_NDX_VARIANT = {1: "FileNotFound", 2: "LineNotFound", 3: "ParseLineErr", 8: "ParseGroupNameErr", 9: "InvalidAtomIndex"}


class ParseGroError(Exception):      # errors.rs: ParseGroError
    def __init__(self, variant, detail):
        super().__init__("ParseGroError::%s(%r)" % (variant, detail)); self.variant, self.detail = variant, detail


class ParseNdxError(Exception):      # errors.rs: ParseNdxError (the two *Warning variants carry sets of names)
    def __init__(self, variant, detail):
        super().__init__("ParseNdxError::%s(%r)" % (variant, detail)); self.variant, self.detail = variant, detail


class Structure:
    """what read_gro returns, as arrays: title, positions [n,3], velocities [n,3] (NaN rows where absent), box9 or None,
    resid / atomid (uint64), resname / atomname (lists of str)"""

    def __init__(self, path):
        lib = _lib.load()
        h = C.c_void_p(); code = C.c_int(0); buf = C.create_string_buffer(512)
        st = lib.gr_gro_read(str(path).encode(), C.byref(h), C.byref(code), buf, 512)
        if st != _lib.OK:
            raise ParseGroError(_GRO_VARIANT.get(code.value, "Unknown"), buf.value.decode(errors="replace"))
        try:
            n = int(lib.gr_structure_n_atoms(h))
            self.title = lib.gr_structure_title(h).decode(errors="replace")
            self.n_atoms = n
            self.positions = np.zeros((n, 3), np.float32); self.velocities = np.zeros((n, 3), np.float32)
            lib.gr_structure_positions(h, self.positions.ctypes.data_as(C.c_void_p))
            lib.gr_structure_velocities(h, self.velocities.ctypes.data_as(C.c_void_p))
            b = np.zeros(9, np.float32)
            self.box9 = b if lib.gr_structure_box(h, b.ctypes.data_as(C.c_void_p)) == _lib.OK else None
            self.resid = np.zeros(n, np.uint64); self.atomid = np.zeros(n, np.uint64); self.resname = []; self.atomname = []
            rn, an = C.create_string_buffer(8), C.create_string_buffer(8)
            ri, ai = C.c_uint64(0), C.c_uint64(0)
            for i in range(n):
                lib.gr_structure_atom(h, i, C.byref(ri), C.byref(ai), rn, an)
                self.resid[i] = ri.value; self.atomid[i] = ai.value
                self.resname.append(rn.value.decode()); self.atomname.append(an.value.decode())
        finally:
            lib.gr_structure_free(h)

    def has_velocities(self):
        return bool(np.isfinite(self.velocities).all()) and self.n_atoms > 0

    def indices_where(self, resname=None, name=None, serial=None, resid=None):
        """the name / resname / serial / resid subset of the selection language: each argument a value, a list of values
        or (for the numbers) an inclusive (first, last) range; all given conditions must hold.  -> sorted atom indices"""
        keep = np.ones(self.n_atoms, bool)
        if resname is not None:
            want = {resname} if isinstance(resname, str) else set(resname)
            keep &= np.array([r in want for r in self.resname], bool)
        if name is not None:
            want = {name} if isinstance(name, str) else set(name)
            keep &= np.array([a in want for a in self.atomname], bool)
        for arr, cond in ((self.atomid, serial), (self.resid, resid)):
            if cond is None:
                continue
            if isinstance(cond, tuple) and len(cond) == 2:
                keep &= (arr >= cond[0]) & (arr <= cond[1])
            else:
                keep &= np.isin(arr, np.atleast_1d(np.asarray(cond, np.uint64)))
        return np.nonzero(keep)[0].astype(np.uint64)


def read_ndx_groups(path, n_atoms):
    """-> [(name, indices uint64 as written: 0-based, file order, duplicates kept)] in file order"""
    lib = _lib.load()
    h = C.c_void_p(); code = C.c_int(0); buf = C.create_string_buffer(512)
    st = lib.gr_ndx_read(str(path).encode(), n_atoms, C.byref(h), C.byref(code), buf, 512)
    if st != _lib.OK:
        d = buf.value.decode(errors="replace")
        raise ParseNdxError(_NDX_VARIANT.get(code.value, "Unknown"), int(d) if code.value == 9 else d)
    try:
        out = []
        for g in range(lib.gr_ndx_n_groups(h)):
            k = lib.gr_ndx_group_size(h, g)
            idx = np.zeros(max(k, 1), np.uint64)
            if k:
                lib.gr_ndx_group_indices(h, g, idx.ctypes.data_as(C.c_void_p))
            out.append((lib.gr_ndx_group_name(h, g).decode(errors="replace"), idx[:k]))
        return out
    finally:
        lib.gr_ndx_free(h)


def system_from_gro(path, masses=None, n_slots=1, device=0):
    """System::from_file for a gro file: atoms, positions, box and the default groups `all` / `All` (system/mod.rs:118-176)"""
    from .system import System
    s = Structure(path)
    sysm = System(s.n_atoms, masses=masses, box=s.box9, positions=s.positions, n_slots=n_slots, device=device, name=s.title)
    sysm.structure = s
    if s.n_atoms:
        sysm.group_create_from_ranges("All", [(0, s.n_atoms - 1)])
    return sysm


def system_read_ndx(system, path):
    """System::read_ndx (ndx_io.rs:31-98): all or nothing; returns normally or raises ParseNdxError -- the two warnings
    (InvalidNamesWarning first, then DuplicateGroupsWarning) are raised AFTER the groups have been created, like the reference"""
    lib = _lib.load()
    h = C.c_void_p(); code = C.c_int(0); buf = C.create_string_buffer(512)
    st = lib.gr_ndx_read(str(path).encode(), system.n_atoms, C.byref(h), C.byref(code), buf, 512)
    if st != _lib.OK:
        d = buf.value.decode(errors="replace")
        raise ParseNdxError(_NDX_VARIANT.get(code.value, "Unknown"), int(d) if code.value == 9 else d)
    try:
        names = [lib.gr_ndx_group_name(h, g).decode(errors="replace") for g in range(lib.gr_ndx_n_groups(h))]
        forbidden = set("'\"&|!@()<>=")
        invalid = {n for n in names if not n.strip() or any(ch in forbidden for ch in n)}
        seen, dup = set(), set()
        for n in names:
            if n in invalid:
                continue
            if n in seen or system.group_exists(n):
                dup.add(n)
            seen.add(n)
        ninv, ndup = C.c_size_t(0), C.c_size_t(0)
        st = lib.gr_ndx_install(h, system._ctx, C.byref(ninv), C.byref(ndup))
        if st != _lib.OK:
            raise RuntimeError("gr_ndx_install: status %d" % st)
    finally:
        lib.gr_ndx_free(h)
    if invalid:
        raise ParseNdxError("InvalidNamesWarning", invalid)
    if dup:
        raise ParseNdxError("DuplicateGroupsWarning", dup)
