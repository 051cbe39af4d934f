"""xtc trajectory reader on top of gr_xtc_* (the library's own decoder, groan_rs_amd/csrc/gr_xtc.h).

Mirrors what the reference's XtcReader gives the path (src/io/xtc_io/mod.rs; TrajRead::update_system,
src/io/traj_read.rs:160-186): frames as (positions[n,3] float32, box9, step, time), random access by index
(the reference's with_range / with_step / per-thread skipping), thread-safe decoding into caller buffers."""
import ctypes as C

import numpy as np

from . import _lib


class XtcError(Exception):
    def __init__(self, status, what):
        super().__init__("%s (status %d)" % (what, status))
        self.status = status


class XtcFile:
    def __init__(self, path):
        self._lib = _lib.load()
        st = C.c_int(0)
        self._x = self._lib.gr_xtc_open(str(path).encode(), C.byref(st))
        if not self._x:
            raise XtcError(st.value, "cannot open %s: %s" % (path, self._lib.gr_status_string(st.value).decode()))
        self.n_atoms = int(self._lib.gr_xtc_n_atoms(self._x))
        self.n_frames = int(self._lib.gr_xtc_n_frames(self._x))

    def close(self):
        if getattr(self, "_x", None):
            self._lib.gr_xtc_close(self._x)
            self._x = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.n_frames

    def frame_info(self, i):
        step = C.c_uint64(0); time = C.c_float(0); prec = C.c_float(0); box = np.zeros(9, np.float32)
        st = self._lib.gr_xtc_frame_info(self._x, i, C.byref(step), C.byref(time), box.ctypes.data_as(C.c_void_p), C.byref(prec))
        if st != _lib.OK:
            raise XtcError(st, "frame_info(%d)" % i)
        return int(step.value), float(time.value), box, float(prec.value)

    def read_frame(self, i, out=None):
        """-> (positions, box9, step, time, precision); `out` may be any C-contiguous float32 [n_atoms,3] buffer (e.g. pinned)"""
        if out is None:
            out = np.empty((self.n_atoms, 3), np.float32)
        step = C.c_uint64(0); time = C.c_float(0); prec = C.c_float(0); box = np.zeros(9, np.float32)
        st = self._lib.gr_xtc_read_frame(self._x, i, out.ctypes.data_as(C.c_void_p), box.ctypes.data_as(C.c_void_p),
                                         C.byref(step), C.byref(time), C.byref(prec))
        if st != _lib.OK:
            raise XtcError(st, "read_frame(%d)" % i)
        return out, box, int(step.value), float(time.value), float(prec.value)

    def read_frame_prefix(self, i, n_prefix):
        """gr_xtc_read_frame_prefix: the first n_prefix atoms of frame i (what a GroupXtcReader needs for a group whose last atom
        is n_prefix - 1, molly_xtc.rs:475-560) -> (positions [n_prefix, 3], box9, step, time, precision, stream_bytes_read)"""
        n_prefix = int(min(n_prefix, self.n_atoms))
        xyz = np.zeros((max(n_prefix, 1), 3), np.float32); box = np.zeros(9, np.float32)
        step = C.c_uint64(0); time = C.c_float(0); prec = C.c_float(0); got = C.c_uint64(0)
        st = self._lib.gr_xtc_read_frame_prefix(self._x, int(i), n_prefix, xyz.ctypes.data_as(C.c_void_p), box.ctypes.data_as(C.c_void_p),
                                                C.byref(step), C.byref(time), C.byref(prec), C.byref(got))
        if st != 0:
            raise XtcError(st, "frame %d" % i)
        return xyz[:n_prefix], box, int(step.value), float(time.value), float(prec.value), int(got.value)

    def read_frames_device(self, system, first_frame, n_frames, first_slot=0, frame_step=1, host_threads=0, group=None):
        """frames first_frame, first_frame + frame_step, ... unpacked ON THE DEVICE into slots first_slot ... (asynchronous);
        group = name: only that group's atoms change (GroupXtcReader / System::group_xtc_iter) -> (steps, times)"""
        steps = np.zeros(n_frames, np.uint64); times = np.zeros(n_frames, np.float32)
        if group is None:
            st = self._lib.gr_xtc_read_frames_device(self._x, int(first_frame), int(n_frames), int(frame_step), system._ctx, int(first_slot), int(host_threads),
                                                     steps.ctypes.data_as(C.c_void_p), times.ctypes.data_as(C.c_void_p))
        else:
            st = self._lib.gr_xtc_read_frames_device_group(self._x, int(first_frame), int(n_frames), int(frame_step), system._ctx, int(first_slot), group.encode(),
                                                           int(host_threads), steps.ctypes.data_as(C.c_void_p), times.ctypes.data_as(C.c_void_p))
        if st != 0:
            raise XtcError(st, system._lib.gr_last_error(system._ctx).decode(errors="replace"))
        return steps, times

    def frames(self, start=0, stop=None, step=1):
        """iterable for TrajReader: (positions, box9, step, time) -- `xtc_iter(..).with_range/with_step` in frame indices"""
        stop = self.n_frames if stop is None else min(stop, self.n_frames)
        for i in range(start, stop, step):
            p, b, s, t, _ = self.read_frame(i)
            yield p, b, s, t

    def __iter__(self):
        return self.frames()


class XtcWriter:
    """XtcWriter (src/io/xtc_io/mod.rs:256-331) over gr_xtc_writer_*: the library's own encoder, byte-compatible with the
    reference's writer.  `write_frame` takes host coordinates, `write_slots` streams device frames out (fitted trajectories)."""

    def __init__(self, path):
        self._lib = _lib.load()
        st = C.c_int(0)
        self._w = self._lib.gr_xtc_writer_open(str(path).encode(), C.byref(st))
        if not self._w:
            raise XtcError(st.value, "cannot create %s" % path)

    def write_frame(self, positions, box9, step=0, time=0.0, precision=1000.0):
        x = np.ascontiguousarray(positions, np.float32)
        b = None if box9 is None else np.ascontiguousarray(box9, np.float32)
        st = self._lib.gr_xtc_write_frame(self._w, x.shape[0], x.ctypes.data_as(C.c_void_p), None if b is None else b.ctypes.data_as(C.c_void_p),
                                          int(step), C.c_float(time), C.c_float(precision))
        if st != _lib.OK:
            raise XtcError(st, "write_frame")

    def write_slots(self, system, first_slot, n_frames, group=None, steps=None, times=None, precision=1000.0, host_threads=0):
        s = None if steps is None else np.ascontiguousarray(steps, np.int64)
        t = None if times is None else np.ascontiguousarray(times, np.float32)
        st = self._lib.gr_xtc_write_slots(self._w, system._ctx, first_slot, n_frames, None if group is None else group.encode(),
                                          None if s is None else s.ctypes.data_as(C.c_void_p), None if t is None else t.ctypes.data_as(C.c_void_p),
                                          C.c_float(precision), host_threads)
        if st != _lib.OK:
            raise XtcError(st, "write_slots: %s" % self._lib.gr_last_error(system._ctx).decode(errors="replace"))

    def close(self):
        if getattr(self, "_w", None):
            st = self._lib.gr_xtc_writer_close(self._w)
            self._w = None
            if st != _lib.OK:
                raise XtcError(st, "close")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
