#!/usr/bin/env python3
"""Expected contents of the trr fixtures, read with the reference's own vendored C xdrfile (oracle/_ref, compiled where it
lies): trr_expected.json = per file the frame count, steps, times, lambdas, box matrices and sha256 digests of the position /
velocity / force arrays (float32, little endian) of every frame.  Run in the build container."""
import ctypes as C
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libxdrfile_ref.so"))
LIB.xdrfile_open.restype = C.c_void_p; LIB.xdrfile_open.argtypes = [C.c_char_p, C.c_char_p]
LIB.xdrfile_close.argtypes = [C.c_void_p]
LIB.read_trr_natoms.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
LIB.read_trr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


def main():
    out = {}
    for name in sorted(f for f in os.listdir(HERE) if f.endswith(".trr")):
        path = os.path.join(HERE, name).encode()
        n = C.c_int(0)
        assert LIB.read_trr_natoms(path, C.byref(n)) == 0
        fh = LIB.xdrfile_open(path, b"r")
        frames = []
        while True:
            x = np.zeros((n.value, 3), np.float32); v = np.zeros_like(x); f = np.zeros_like(x); m = np.zeros((3, 3), np.float32)
            step = C.c_int(0); t = C.c_float(0); lam = C.c_float(0)
            if LIB.read_trr(fh, n.value, C.byref(step), C.byref(t), C.byref(lam), m.ctypes.data, x.ctypes.data, v.ctypes.data, f.ctypes.data) != 0:
                break
            frames.append({"step": step.value, "time": t.value, "lambda": lam.value, "box_rowmajor": m.reshape(-1).tolist(),
                           "x": hashlib.sha256(x.tobytes()).hexdigest(), "v": hashlib.sha256(v.tobytes()).hexdigest(), "f": hashlib.sha256(f.tobytes()).hexdigest(),
                           "x_first": x[0].tolist(), "has_x": bool(x.any()), "has_v": bool(v.any()), "has_f": bool(f.any())})
        LIB.xdrfile_close(fh)
        out[name] = {"n_atoms": n.value, "frames": frames}
        print(name, n.value, "atoms", len(frames), "frames")
    json.dump(out, open(os.path.join(HERE, "trr_expected.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
