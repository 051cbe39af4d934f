"""Synthetic xtc test trajectories shared by the CPU and the GPU suites, and the sha256 pins of the files they encode to.

The reference's own xtc writer (its vendored C xdrfile, oracle/_ref) exists in the build container only: there
tests/golden/make_xtc_pins.py writes every case with it and records the sha256 of the file (tests/golden/xtc_pins.json), and
tests/test_xtc_writer.py proves the library's own encoder byte-identical to it.  On the GPU box the -m gpu tests write the
same cases with the library's OWN encoder and check the file against the pin -- the reference's compiled code never travels.
Everything here is elementwise numpy on seeded generators (no BLAS call), so the coordinates are the same bits on every host."""
import hashlib
import json
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = os.path.join(HERE, "golden", "xtc_pins.json")
BOX = np.array([[30, 0, 0], [0, 30, 0], [10, 10, 25]], np.float32)
CASES = ["tiny9", "water", "gas", "wide_range", "wide_product", "high_precision", "mixed", "big_water"]


def water_like(rng, n, span):
    """molecules of 3 atoms within 0.1 nm of each other -> long runs of 'small' deltas, plus scattered ions"""
    nm = n // 3
    o = rng.uniform(0, span, (nm, 3))
    x = np.repeat(o, 3, axis=0) + rng.normal(0, 0.06, (nm * 3, 3))
    extra = rng.uniform(0, span, (n - nm * 3, 3))
    return np.concatenate([x, extra]).astype(np.float32)


def branch_case(case):
    """-> (frames, 3x3 box, precision): coordinates that hit one branch of the format each (<= 9 atoms raw, water-like runs,
    wide ranges -> fixed-width fields, > 64-bit packing, high precision, a 5e5-atom frame of BASELINE config 5's size)"""
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    prec = 1000.0
    if case == "tiny9": frames = [rng.uniform(0, 5, (9, 3)).astype(np.float32)] * 3
    elif case == "water": frames = [water_like(rng, 30000, 20.0) for _ in range(3)]
    elif case == "gas": frames = [rng.uniform(-50, 50, (5000, 3)).astype(np.float32) for _ in range(3)]
    elif case == "wide_range": frames = [np.concatenate([rng.uniform(0, 10, (4000, 3)), [[20000.0, 3.0, 4.0]]]).astype(np.float32) for _ in range(2)]
    elif case == "wide_product": frames = [rng.uniform(0, 8000, (3000, 3)).astype(np.float32) for _ in range(2)]
    elif case == "high_precision":
        prec = 100000.0
        frames = [water_like(rng, 3000, 6.0) for _ in range(3)]
    elif case == "big_water": frames = [water_like(rng, 500_000, 17.0) for _ in range(2)]
    elif case == "mixed": frames = [np.concatenate([water_like(rng, 9000, 12.0), rng.uniform(0, 12, (1000, 3)).astype(np.float32), water_like(rng, 2001, 3.0)]) for _ in range(4)]
    else: raise KeyError(case)
    return frames, BOX, prec


def frac_to_cart(f, boxm):
    """rows of fractional coordinates times the (lower-triangular, row-vector) box matrix, elementwise in float64"""
    b = np.asarray(boxm, np.float64)
    return f[:, 0:1] * b[0][None, :] + f[:, 1:2] * b[1][None, :] + f[:, 2:3] * b[2][None, :]


def octahedron_case(n=500_000, n_distinct=4):
    """BASELINE config 5's shape: a water-like truncated octahedron (simbox.rs:329-342) of n atoms with a compact 30 000-atom
    'solute' that drifts through the periodic boundary; -> (frames (wrapped by the caller), box9, 3x3 box)"""
    import oracle_lib as O
    box9 = O.box_from_lengths_angles([18.0, 18.0, 18.0], [70.53, 109.47, 70.53])
    boxm = np.array([[box9[0], 0, 0], [box9[5], box9[1], 0], [box9[7], box9[8], box9[2]]], np.float32)
    rng = np.random.default_rng(5)
    nm = n // 3
    mol = frac_to_cart(rng.uniform(0, 1, (nm, 3)), boxm)
    base = np.repeat(mol, 3, axis=0) + rng.normal(0, 0.06, (nm * 3, 3))
    base = np.concatenate([base, frac_to_cart(rng.uniform(0, 1, (n - 3 * nm, 3)), boxm)])
    base[0:30_000] = O.box_center(box9) + rng.normal(0, 1.2, (30_000, 3))
    frames = []
    for f in range(n_distinct):
        fr = base + rng.normal(0, 0.02, base.shape)
        fr[0:30_000] += rng.uniform(-6, 6, 3)
        frames.append(O.wrap_atoms(fr.astype(np.float32), np.arange(n), box9))
    return frames, box9, boxm


def box9_of(boxm):
    m = np.asarray(boxm, np.float32)
    return np.array([m[0, 0], m[1, 1], m[2, 2], m[0, 1], m[0, 2], m[1, 0], m[1, 2], m[2, 0], m[2, 1]], np.float32)


def write_own(G, path, frames, boxm, precision):
    """the library's own encoder, with the step / time sequence the pins were made with (step 10 i, time 0.5 i)"""
    b9 = box9_of(boxm)
    with G.XtcWriter(path) as w:
        for i, x in enumerate(frames):
            w.write_frame(x, b9, step=i * 10, time=i * 0.5, precision=precision)


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def pin(name):
    return json.load(open(PINS))[name]
