"""The library's own xtc decoder (groan_rs_amd/csrc/gr_xtc.h, host code) -- bit-exact against
  (1) committed data files of the reference's test suite with their expected decodes (tests/golden/xtc_expected.json
      = sha256 of the float stream the reference's vendored xdrfile produces, plus steps/times/boxes; tric_small.npz /
      short_traj.npz hold decoded coordinates), and
  (2) when oracle/_ref is built: files written by the reference's xdrfile writer from synthetic coordinates that hit
      every branch of the format (<= 9 atoms raw, water-like runs, wide ranges -> fixed-width fields, > 64-bit packing).
CPU only (the decoder is host code)."""
import ctypes as C
import hashlib
import json
import os
import threading

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF_SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libxdrfile_ref.so")


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def box9_from_rowmajor(m):
    return np.array([m[0], m[4], m[8], m[1], m[2], m[3], m[5], m[6], m[7]], np.float32)


@pytest.mark.parametrize("name", ["triclinic_trajectory.xtc", "octahedron_trajectory.xtc", "dodecahedron_trajectory.xtc", "short_trajectory.xtc"])
def test_reference_data_files_bit_exact(G, name, tric_small, short_traj):
    exp = json.load(open(os.path.join(GOLD, "xtc_expected.json")))[name]
    x = G.XtcFile(os.path.join(GOLD, name))
    assert x.n_frames == exp["n_frames"] and x.n_atoms == exp["n_atoms"]
    frames = []
    for i in range(x.n_frames):
        pos, box, step, time, prec = x.read_frame(i)
        assert step == exp["steps"][i] and np.float32(time) == np.float32(exp["times"][i]) and np.float32(prec) == np.float32(exp["precision"])
        assert np.array_equal(box, box9_from_rowmajor(exp["boxes_rowmajor"][i]))
        assert x.frame_info(i)[0] == step
        frames.append(pos)
    allf = np.stack(frames)
    assert hashlib.sha256(allf.astype("<f4").tobytes()).hexdigest() == exp["sha256_coords_f32le"]
    if name == "short_trajectory.xtc":
        assert np.array_equal(allf[:, short_traj["keep"].astype(np.int64)], short_traj["frames"])
    else:
        assert np.array_equal(allf, tric_small[name.replace("_trajectory.xtc", "_frames")])
    # iteration protocol used by TrajReader, with start/stop/step
    got = [s for _, _, s, _ in x.frames(1, None, 3)]
    assert got == exp["steps"][1::3]
    x.close()


def test_errors(G, tmp_path):
    with pytest.raises(G.XtcError) as e:
        G.XtcFile(tmp_path / "nope.xtc")
    assert e.value.status == G._lib.E_IO
    bad = tmp_path / "bad.xtc"
    bad.write_bytes(b"\x00\x00\x00\x01" + b"\x00" * 60)
    with pytest.raises(G.XtcError) as e:
        G.XtcFile(bad)
    assert e.value.status == G._lib.E_FORMAT
    src = open(os.path.join(GOLD, "triclinic_trajectory.xtc"), "rb").read()
    trunc = tmp_path / "trunc.xtc"
    trunc.write_bytes(src[: len(src) - 37])
    with pytest.raises(G.XtcError):
        G.XtcFile(trunc)
    x = G.XtcFile(os.path.join(GOLD, "triclinic_trajectory.xtc"))
    with pytest.raises(G.XtcError) as e:
        x.read_frame(10 ** 6)
    assert e.value.status == G._lib.E_OUT_OF_RANGE


# ------------------------------------------------------------------ against the reference's own writer/reader (oracle/_ref)
def _ref():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    lib = C.CDLL(REF_SO)
    lib.xdrfile_open.restype = C.c_void_p
    lib.xdrfile_open.argtypes = [C.c_char_p, C.c_char_p]
    lib.xdrfile_close.argtypes = [C.c_void_p]
    lib.write_xtc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_float]
    lib.read_xtc.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    return lib


def write_with_ref(path, frames, box, precision):
    lib = _ref()
    fh = lib.xdrfile_open(str(path).encode(), b"w")
    assert fh
    m = np.ascontiguousarray(box, np.float32)
    for i, x in enumerate(frames):
        x = np.ascontiguousarray(x, np.float32)
        assert lib.write_xtc(fh, x.shape[0], i * 10, C.c_float(i * 0.5), m.ctypes.data, x.ctypes.data, C.c_float(precision)) == 0
    lib.xdrfile_close(fh)


def read_with_ref(path, n):
    lib = _ref()
    fh = lib.xdrfile_open(str(path).encode(), b"r")
    out = []
    while True:
        x = np.zeros((n, 3), np.float32); m = np.zeros((3, 3), np.float32)
        step = C.c_int(0); t = C.c_float(0); p = C.c_float(0)
        if lib.read_xtc(fh, n, C.byref(step), C.byref(t), m.ctypes.data, x.ctypes.data, C.byref(p)) != 0:
            break
        out.append(x)
    lib.xdrfile_close(fh)
    return np.stack(out)


def water_like(rng, n, span):
    """molecules of 3 atoms within 0.1 nm of each other -> long runs of 'small' deltas, plus scattered ions"""
    nm = n // 3
    o = rng.uniform(0, span, (nm, 3))
    x = np.repeat(o, 3, axis=0) + rng.normal(0, 0.06, (nm * 3, 3))
    extra = rng.uniform(0, span, (n - nm * 3, 3))
    return np.concatenate([x, extra]).astype(np.float32)


@pytest.mark.parametrize("case", ["tiny9", "water", "gas", "wide_range", "wide_product", "high_precision", "mixed"])
def test_against_reference_writer(G, tmp_path, case):
    import zlib
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    box = np.array([[30, 0, 0], [0, 30, 0], [10, 10, 25]], np.float32)
    prec = 1000.0
    if case == "tiny9":
        frames = [rng.uniform(0, 5, (k, 3)).astype(np.float32) for k in (9,)] * 3
    elif case == "water":
        frames = [water_like(rng, 30000, 20.0) for _ in range(3)]
    elif case == "gas":
        frames = [rng.uniform(-50, 50, (5000, 3)).astype(np.float32) for _ in range(3)]
    elif case == "wide_range":      # a range above 2^24 quanta -> three fixed-width fields instead of the mixed radix
        frames = [np.concatenate([rng.uniform(0, 10, (4000, 3)), [[20000.0, 3.0, 4.0]]]).astype(np.float32) for _ in range(2)]
    elif case == "wide_product":    # product of the three ranges above 2^64 -> 128-bit unpacking
        frames = [rng.uniform(0, 8000, (3000, 3)).astype(np.float32) for _ in range(2)]
    elif case == "high_precision":
        prec = 100000.0
        frames = [water_like(rng, 3000, 6.0) for _ in range(3)]
    else:
        frames = [np.concatenate([water_like(rng, 9000, 12.0), rng.uniform(0, 12, (1000, 3)).astype(np.float32), water_like(rng, 2001, 3.0)]) for _ in range(4)]
    path = tmp_path / (case + ".xtc")
    write_with_ref(path, frames, box, prec)
    n = frames[0].shape[0]
    want = read_with_ref(path, n)
    x = G.XtcFile(path)
    assert x.n_atoms == n and x.n_frames == len(frames)
    for i in range(x.n_frames):
        pos, b9, step, time, p = x.read_frame(i)
        assert np.array_equal(pos, want[i]), (case, i, np.abs(pos - want[i]).max())
        assert step == i * 10 and time == np.float32(i * 0.5) and (n <= 9 or p == np.float32(prec))
        assert np.array_equal(b9, np.array([30, 30, 25, 0, 0, 0, 0, 10, 10], np.float32))
    x.close()


def test_threaded_decode_is_consistent(G):
    """one frame per thread into caller buffers (the decode stage of the upload pipeline)"""
    x = G.XtcFile(os.path.join(GOLD, "short_trajectory.xtc"))
    serial = [x.read_frame(i)[0].copy() for i in range(x.n_frames)]
    out = [np.empty((x.n_atoms, 3), np.float32) for _ in range(x.n_frames)]
    ths = [threading.Thread(target=lambda i=i: x.read_frame(i, out=out[i])) for i in range(x.n_frames)]
    [t.start() for t in ths]; [t.join() for t in ths]
    for a, b in zip(serial, out):
        assert np.array_equal(a, b)
    x.close()


@pytest.mark.parametrize("name", ["short_trajectory.xtc", "triclinic_trajectory.xtc", "octahedron_trajectory.xtc"])
def test_prefix_decode_is_the_full_decodes_prefix(G, name):
    """Partial-frame reading (GroupXtcReader, molly_xtc.rs:475-560): the first n atoms decoded on their own are bit for bit the
    first n atoms of the full decode, for prefixes that end inside runs, on run boundaries, at 0 and at the whole frame -- and a
    short prefix reads only the head of the frame's bit stream."""
    x = G.XtcFile(os.path.join(GOLD, name))
    n = x.n_atoms
    for f in (0, x.n_frames - 1):
        full, box, step, time, prec = x.read_frame(f)
        for k in sorted({0, 1, 2, 3, 7, 8, 9, 10, 31, 32, 33, 61, 363, n // 3, n - 1, n, n + 5} & set(range(0, n + 6))):
            part, pbox, pstep, ptime, pprec, got = x.read_frame_prefix(f, k)
            kk = min(k, n)
            assert part.shape == (kk, 3) and np.array_equal(part.view(np.uint32), full[:kk].view(np.uint32)), (name, f, k)
            assert np.array_equal(pbox, box) and (pstep, ptime, pprec) == (step, time, prec)
        if n > 5000:
            _, _, _, _, _, got_small = x.read_frame_prefix(f, 61)
            _, _, _, _, _, got_all = x.read_frame_prefix(f, n)
            assert got_small < got_all / 20                       # 61 of 16844 atoms: a few hundred bytes instead of ~60 kB
    x.close()


def test_prefix_decode_on_every_format_branch(G, tmp_path):
    """the same on synthetic systems written with the library's encoder (water runs, wide ranges, high precision): random prefixes"""
    rng = np.random.default_rng(77)
    for case, (frames, prec) in {"water": ([water_like(rng, 30000, 20.0) for _ in range(2)], 1000.0),
                                 "gas": ([rng.uniform(-50, 50, (5000, 3)).astype(np.float32) for _ in range(2)], 1000.0),
                                 "wide": ([np.concatenate([rng.uniform(0, 10, (4000, 3)), [[20000.0, 3.0, 4.0]]]).astype(np.float32)], 1000.0),
                                 "fine": ([water_like(rng, 3000, 6.0)], 100000.0)}.items():
        path = tmp_path / (case + ".xtc")
        with G.XtcWriter(path) as w:
            for k, fr in enumerate(frames):
                w.write_frame(fr, [30, 30, 25, 0, 0, 0, 0, 10, 10], step=k, time=float(k), precision=prec)
        x = G.XtcFile(path)
        for f in range(x.n_frames):
            full = x.read_frame(f)[0]
            for k in [int(v) for v in rng.integers(0, x.n_atoms + 1, 12)] + [363]:
                part = x.read_frame_prefix(f, k)[0]
                assert np.array_equal(part.view(np.uint32), full[:min(k, x.n_atoms)].view(np.uint32)), (case, f, k)
        x.close()
