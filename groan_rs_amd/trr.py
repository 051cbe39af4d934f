"""trr trajectories: the Python mirror of the library's own reader (include/groan_hip.h `gr_trr_*`), standing in for the
reference's TrrReader (src/io/trr_io.rs:30-135, 143-260).  Frames are random-access; `frames()` feeds TrajReader with
(positions, box9, step, time) like XtcFile.frames, with the reference's rule that an all-zero position means "no position"."""
import ctypes as C

import numpy as np

from . import _lib
from .xtc import XtcError


class TrrFile:
    def __init__(self, path):
        self._lib = _lib.load()
        st = C.c_int(0)
        self._t = self._lib.gr_trr_open(str(path).encode(), C.byref(st))
        if not self._t:
            raise XtcError(st.value, "cannot open %s: %s" % (path, self._lib.gr_status_string(st.value).decode()))
        self.n_atoms = int(self._lib.gr_trr_n_atoms(self._t))
        self.n_frames = int(self._lib.gr_trr_n_frames(self._t))

    def close(self):
        if getattr(self, "_t", None):
            self._lib.gr_trr_close(self._t)
            self._t = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def __del__(self): self.close()

    def frame_info(self, frame):
        """-> dict(step, time, lambda_, box9 or None, positions / velocities / forces present, double_precision)"""
        step = C.c_uint64(0); t = C.c_float(0); lam = C.c_float(0); sec = C.c_int(0); dbl = C.c_int(0)
        box = np.zeros(9, np.float32)
        st = self._lib.gr_trr_frame_info(self._t, frame, C.byref(step), C.byref(t), C.byref(lam), box.ctypes.data_as(C.c_void_p), C.byref(sec), C.byref(dbl))
        if st != _lib.OK:
            raise XtcError(st, "frame_info(%d)" % frame)
        return {"step": int(step.value), "time": float(t.value), "lambda_": float(lam.value), "box9": box if sec.value & 8 else None,
                "positions": bool(sec.value & 1), "velocities": bool(sec.value & 2), "forces": bool(sec.value & 4), "double_precision": bool(dbl.value)}

    def read_frame(self, frame, velocities=False, forces=False, out=None):
        """-> (positions [n,3], velocities or None, forces or None, box9 or None, step, time, lambda); absent sections are zeros"""
        n = self.n_atoms
        x = out if out is not None else np.zeros((n, 3), np.float32)
        v = np.zeros((n, 3), np.float32) if velocities else None
        f = np.zeros((n, 3), np.float32) if forces else None
        step = C.c_uint64(0); t = C.c_float(0); lam = C.c_float(0)
        box = np.zeros(9, np.float32)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        st = self._lib.gr_trr_read_frame(self._t, frame, p(x), p(v), p(f), p(box), C.byref(step), C.byref(t), C.byref(lam))
        if st != _lib.OK:
            raise XtcError(st, "read_frame(%d)" % frame)
        info = self.frame_info(frame)
        return x, v, f, info["box9"], int(step.value), float(t.value), float(lam.value)

    def read_frames_device(self, system, first_frame, n_frames, first_slot=0, frame_step=1):
        """frames -> slots of `system`: the raw big-endian positions cross PCIe and are converted on the GPU (asynchronous);
        an all-zero position arrives as the missing-position marker.  -> (steps uint64[n], times float32[n])"""
        steps = np.zeros(n_frames, np.uint64); times = np.zeros(n_frames, np.float32)
        st = self._lib.gr_trr_read_frames_device(self._t, first_frame, n_frames, frame_step, system._ctx, first_slot,
                                                 steps.ctypes.data_as(C.c_void_p), times.ctypes.data_as(C.c_void_p))
        if st != _lib.OK:
            raise XtcError(st, "read_frames_device(%d, %d): %s" % (first_frame, n_frames, self._lib.gr_last_error(system._ctx).decode(errors="replace")))
        return steps, times

    def frames(self, start=0, stop=None, step=1):
        """iterable for TrajReader: (positions, box9, step, time); a zero position is the reference's "no position" (NaN in x)"""
        stop = self.n_frames if stop is None else min(stop, self.n_frames)
        for fr in range(start, stop, step):
            x, _, _, box, s, t, _ = self.read_frame(fr)
            zero = ~x.any(axis=1)
            if zero.any():
                x[zero, 0] = np.nan
            yield x, box, s, t


class TrrWriter:
    """TrrWriter (src/io/trr_io.rs:441-520): single-precision frames with box, positions, velocities and forces -- like the
    reference, every frame carries all three arrays, zeros where the system has none; byte-compatible with its writer."""

    def __init__(self, path):
        self._lib = _lib.load()
        st = C.c_int(0)
        self._w = self._lib.gr_trr_writer_open(str(path).encode(), C.byref(st))
        if not self._w:
            raise XtcError(st.value, "cannot create %s" % path)

    def write_frame(self, positions, box9, step=0, time=0.0, lambda_=0.0, velocities=None, forces=None):
        x = np.ascontiguousarray(positions, np.float32)
        v = np.zeros_like(x) if velocities is None else np.ascontiguousarray(velocities, np.float32)
        f = np.zeros_like(x) if forces is None else np.ascontiguousarray(forces, np.float32)
        b = None if box9 is None else np.ascontiguousarray(box9, np.float32)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        st = self._lib.gr_trr_write_frame(self._w, x.shape[0], p(x), p(v), p(f), p(b), int(step), C.c_float(time), C.c_float(lambda_))
        if st != _lib.OK:
            raise XtcError(st, "write_frame")

    def close(self):
        if getattr(self, "_w", None):
            st = self._lib.gr_trr_writer_close(self._w)
            self._w = None
            if st != _lib.OK:
                raise XtcError(st, "close")

    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def __del__(self):
        try: self.close()
        except Exception: pass
