"""The library's own xtc ENCODER (groan_rs_amd/csrc/gr_xtc.h, host code) must write byte for byte what the reference's
writer writes (XtcWriter over xdrfile's write_xtc, src/io/xtc_io/mod.rs:256-331) -- the reference's golden fitted
trajectories are compared as files.
  (1) known answers: the committed data files of the reference's test suite.  Decoding is exact (test_xtc_decoder.py), and
      re-quantising a decoded coordinate returns its integer, so encode(decode(file)) has to reproduce the file itself;
  (2) when oracle/_ref is built: synthetic coordinates through the reference's writer and through ours, same bytes, over
      every branch of the format (raw <= 9 atoms, water runs, wide ranges, > 64-bit packing, precision 1e5, negative coordinates).
CPU only (host code)."""
import os
import zlib

import numpy as np
import pytest

from test_xtc_decoder import GOLD, REF_SO, read_with_ref, water_like, write_with_ref


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def rows_to_box9(m):
    m = np.asarray(m, np.float32).ravel()
    return np.array([m[0], m[4], m[8], m[1], m[2], m[3], m[5], m[6], m[7]], np.float32)


@pytest.mark.parametrize("name", ["triclinic_trajectory.xtc", "octahedron_trajectory.xtc", "dodecahedron_trajectory.xtc", "short_trajectory.xtc"])
def test_reencoding_the_reference_files_reproduces_them(G, tmp_path, name):
    src = os.path.join(GOLD, name)
    x = G.XtcFile(src)
    out = tmp_path / name
    with G.XtcWriter(out) as w:
        for i in range(x.n_frames):
            pos, box, step, time, prec = x.read_frame(i)
            w.write_frame(pos, box, step=step, time=time, precision=prec)
    x.close()
    assert open(out, "rb").read() == open(src, "rb").read()


@pytest.mark.parametrize("case", ["tiny9", "water", "gas", "negative", "wide_range", "wide_product", "high_precision", "mixed", "no_box_nan"])
def test_same_bytes_as_the_reference_writer(G, tmp_path, case):
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(zlib.crc32(case.encode()) + 1)
    boxm = np.array([[30, 0, 0], [0, 30, 0], [10, 10, 25]], np.float32)
    prec = 1000.0
    if case == "tiny9": frames = [rng.uniform(-5, 5, (9, 3)).astype(np.float32) for _ in range(3)]
    elif case == "water": frames = [water_like(rng, 30000, 20.0) for _ in range(3)]
    elif case == "gas": frames = [rng.uniform(0, 50, (5000, 3)).astype(np.float32) for _ in range(3)]
    elif case == "negative": frames = [(water_like(rng, 6000, 8.0) - 4.0).astype(np.float32) for _ in range(3)]
    elif case == "wide_range": frames = [np.concatenate([rng.uniform(0, 10, (4000, 3)), [[20000.0, 3.0, 4.0]]]).astype(np.float32) for _ in range(2)]
    elif case == "wide_product": frames = [rng.uniform(0, 8000, (3000, 3)).astype(np.float32) for _ in range(2)]
    elif case == "high_precision":
        prec = 100000.0
        frames = [water_like(rng, 3000, 6.0) for _ in range(3)]
    elif case == "no_box_nan":
        frames = [water_like(rng, 999, 5.0) for _ in range(2)]
        boxm = np.zeros((3, 3), np.float32)
    else:
        frames = [np.concatenate([water_like(rng, 9000, 12.0), rng.uniform(0, 12, (1000, 3)).astype(np.float32), water_like(rng, 2001, 3.0)]) for _ in range(4)]
    ref_path, our_path = tmp_path / "ref.xtc", tmp_path / "ours.xtc"
    write_with_ref(ref_path, frames, boxm, prec)
    with G.XtcWriter(our_path) as w:
        for i, f in enumerate(frames):
            g = f.copy()
            if case == "no_box_nan":
                g[5, 0] = np.nan                      # an atom without position is written as the origin
                frames[i][5] = 0.0
            w.write_frame(g, None if case == "no_box_nan" else rows_to_box9(boxm), step=i * 10, time=i * 0.5, precision=prec)
    if case == "no_box_nan":
        write_with_ref(ref_path, frames, boxm, prec)
    a, b = open(our_path, "rb").read(), open(ref_path, "rb").read()
    assert len(a) == len(b) and a == b, (case, len(a), len(b), next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), None))


def test_errors(G, tmp_path):
    with pytest.raises(G.XtcError):
        G.XtcWriter(tmp_path / "no_such_dir" / "x.xtc")


def test_coordinates_the_format_cannot_hold_are_refused(G, tmp_path):
    """x * precision beyond the 32-bit integers of the format (xdrfile.c:1025-1030 prints "Internal overflow compressing
    coordinates." and converts anyway -- undefined behaviour): the frame is refused and nothing reaches the file; a NaN in x
    is the reference's missing position (written as the origin, xtc_io/mod.rs:296-301), a NaN in y or z is not a value"""
    rng = np.random.default_rng(4)
    pos = rng.uniform(0, 5, (40, 3)).astype(np.float32)
    box = np.array([5, 5, 5, 0, 0, 0, 0, 0, 0], np.float32)
    path = tmp_path / "refuse.xtc"
    with G.XtcWriter(path) as w:
        w.write_frame(pos, box, step=1)
        for bad, prec in ((3e6, 1000.0), (-3e6, 1000.0), (np.inf, 1000.0), (3000.0, 1e6)):
            p = pos.copy(); p[7, 1] = bad
            with pytest.raises(G.XtcError) as e:
                w.write_frame(p, box, step=2, precision=prec)
            assert e.value.status == G._lib.GR_E_OUT_OF_RANGE if hasattr(G._lib, "GR_E_OUT_OF_RANGE") else e.value.status == 9
        p = pos.copy(); p[3, 2] = np.nan
        with pytest.raises(G.XtcError):
            w.write_frame(p, box, step=3)
        p = pos.copy(); p[3, 0] = np.nan                      # missing position: fine
        w.write_frame(p, box, step=4)
        wide = pos.copy(); wide[0, 0] = -1.2e6; wide[1, 0] = 1.2e6   # each value fits, their range does not
        with pytest.raises(G.XtcError):
            w.write_frame(wide, box, step=5)
    x = G.XtcFile(path)
    assert x.n_frames == 2                                     # the refused frames left no bytes behind
    got = x.read_frame(1)[0]
    assert np.all(got[3] == 0.0) and np.abs(got[4] - pos[4]).max() <= 0.00051
    x.close()


@pytest.mark.parametrize("seed", range(48))
def test_random_systems_encode_to_the_reference_writers_bytes(G, tmp_path, seed):
    """randomised byte-for-byte comparison with the reference's writer (oracle/_ref = the reference's vendored xdrfile compiled
    where it lies): atom counts around the format's thresholds, coincident and collinear atoms (zero minimum step), identical
    frames, runs of every length, molecules of 2-9 atoms, coordinates on the quantisation half-steps, mixed magnitudes,
    several precisions"""
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.choice([10, 11, 12, 13, 27, 100, 333, 1000, 2501]))
    prec = float(rng.choice([1000.0, 100.0, 10.0, 10000.0, 500.0]))
    span = float(rng.choice([0.5, 3.0, 12.0, 80.0, 900.0]))
    mol = int(rng.integers(1, 10))
    frames = []
    for f in range(3):
        base = rng.uniform(-span if seed % 3 == 0 else 0.0, span, ((n + mol - 1) // mol, 3))
        x = np.repeat(base, mol, axis=0)[:n] + rng.normal(0, float(rng.choice([0.0, 0.002, 0.05, 0.3])), (n, 3))
        kind = rng.integers(0, 6, n)
        x[kind == 1] = x[0]                                                   # coincident atoms
        x[kind == 2] = np.round(x[kind == 2] * prec) / prec + 0.5 / prec      # on the rounding half-steps
        x[kind == 3, 1:] = x[0, 1:]                                           # collinear along x
        if seed % 4 == 1 and f == 2:
            x = frames[-1].astype(np.float64)                                 # an identical frame
        frames.append(x.astype(np.float32))
    boxm = np.array([[span, 0, 0], [0, span, 0], [0, 0, span]], np.float32)
    ref_path, our_path = tmp_path / "ref.xtc", tmp_path / "ours.xtc"
    write_with_ref(ref_path, frames, boxm, prec)
    with G.XtcWriter(our_path) as w:
        for i, fr in enumerate(frames):
            w.write_frame(fr, rows_to_box9(boxm), step=i * 10, time=i * 0.5, precision=prec)
    a, b = open(our_path, "rb").read(), open(ref_path, "rb").read()
    assert len(a) == len(b) and a == b, (seed, n, prec, span, mol, len(a), len(b), next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), None))
    x = G.XtcFile(our_path)                                                  # and our decoder reads it back like the reference's
    want = read_with_ref(ref_path, n)
    for i in range(3):
        assert np.array_equal(x.read_frame(i)[0], want[i])
    x.close()


# ------------------------------------------------------------------ the pins the GPU suite relies on (tests/xtc_cases.py)
import xtc_cases as XC   # noqa: E402


@pytest.mark.parametrize("case", XC.CASES + ["octahedron_5e5_x32"])
def test_pinned_cases_own_encoder_and_reference_writer(G, tmp_path, case):
    """tests/golden/xtc_pins.json holds the sha256 of each synthetic trajectory as the REFERENCE's writer encodes it (made by
    tests/golden/make_xtc_pins.py).  Here: the library's own encoder produces exactly those files -- which is what lets the -m gpu
    tests write their inputs without the reference's compiled code -- and, where oracle/_ref exists, so does the reference."""
    if case == "octahedron_5e5_x32":
        fr, _, box = XC.octahedron_case()
        frames, prec = [fr[f % 4] for f in range(32)], 1000.0
    else:
        frames, box, prec = XC.branch_case(case)
    ours = tmp_path / "ours.xtc"
    XC.write_own(G, ours, frames, box, prec)
    assert XC.sha256_file(ours) == XC.pin(case)["sha256"] and os.path.getsize(ours) == XC.pin(case)["bytes"]
    if os.path.exists(REF_SO) and case != "octahedron_5e5_x32":
        refp = tmp_path / "ref.xtc"
        write_with_ref(refp, frames, box, prec)
        assert XC.sha256_file(refp) == XC.pin(case)["sha256"]
