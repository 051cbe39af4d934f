"""Synthetic workloads of SURVEY.md section 8(d) / BASELINE.json configs, shared by bench.py, the tools and the
full-size parity tests so that "the benchmark's workload" is one definition.

SimBox::from_lengths_angles is the reference's formula (src/structures/simbox.rs:96-123) in f32, operation by
operation; the box kinds are the reference's own test boxes (simbox.rs:300-314 rhombic dodecahedron, :329-342
truncated octahedron)."""
import math

import numpy as np

SEED = 20260424
_f = np.float32


def box_from_lengths_angles(lengths, angles):
    """SimBox::from_lengths_angles (simbox.rs:96-123) -> gro-order box9 (simbox.rs:13-26)."""
    l = [_f(x) for x in lengths]
    a = [_f(x) for x in angles]
    box = np.zeros(9, np.float32)
    box[0] = l[0]
    if a[0] == 90.0 and a[1] == 90.0 and a[2] == 90.0:
        box[1], box[2] = l[1], l[2]
        return box
    pi = _f(math.pi)
    alpha, beta, gamma = [x * pi / _f(180.0) for x in a]
    cos = lambda x: _f(math.cos(x))   # f32 argument, f32 result (libm cosf agrees with the rounded double value to < 1 ulp here;
    sin = lambda x: _f(math.sin(x))   # the full-size tests take their box from this function on both sides)
    box[5] = l[1] * cos(gamma)                                        # v2x
    box[1] = l[1] * sin(gamma)                                        # v2y
    box[7] = l[2] * cos(beta)                                         # v3x
    box[8] = l[2] * (cos(alpha) - cos(beta) * cos(gamma)) / sin(gamma)  # v3y
    box[2] = _f(math.sqrt(l[2] * l[2] - box[7] * box[7] - box[8] * box[8]))   # v3z
    return box


def masses_cycle(n):
    """{H, C, N, O}[i mod 4]"""
    return np.array([1.008, 12.011, 14.007, 15.999], np.float32)[np.arange(n) % 4]


def c4_box(d=24.18):
    """BASELINE configs[3]: rhombic dodecahedron, lengths d,d,d, angles 60,60,90."""
    return box_from_lengths_angles([d, d, d], [60.0, 60.0, 90.0])


def c3_box():
    """BASELINE configs[2]: general triclinic cell 24 x 23 x 22 nm, angles 75 / 80 / 70."""
    return box_from_lengths_angles([24.0, 23.0, 22.0], [75.0, 80.0, 70.0])


def c5_box(d=24.0):
    """BASELINE configs[4]: truncated octahedron (angles 70.53, 109.47, 70.53)."""
    return box_from_lengths_angles([d, d, d], [70.53, 109.47, 70.53])


def blob_radius(box9, frac=0.2):
    """radius of the RMSD-fit blob: 0.2 x the shortest box height (the diagonal of the lower-triangular box matrix)"""
    return frac * float(min(box9[0], box9[1], box9[2]))


def box_matrix(box9):
    """rows = the box vectors a, b, c (gro order -> lower-triangular matrix), float64"""
    b = np.asarray(box9, np.float64)
    return np.array([[b[0], 0.0, 0.0], [b[5], b[1], 0.0], [b[7], b[8], b[2]]])


def wrap_into_cell(pos, box9):
    """positions -> the rectangular cell 0 <= x < ax, 0 <= y < by, 0 <= z < cz (shifts along c, then b, then a), float32.
    Input preparation for synthetic frames: any periodic image is as good as another to the path under test."""
    L = box_matrix(box9)
    p = np.array(pos, np.float64)
    for axis in (2, 1, 0):
        k = np.floor(p[:, axis] / L[axis, axis])
        p -= k[:, None] * L[axis][None, :]
    return p.astype(np.float32)


def proof_failing_frame(ref_pos, box9, kind, seed, noise=0.04):
    """A frame the one-pass RMSD paths must HAND BACK to the literal multi-pass path: the group is wider than half the cell, so no
    single choice of periodic images about its first atom can be proven to be the reference's (DESIGN.md "Image proof"), while
    every atom stays well inside the minimum-image cell about the group's centre, so the reference's own answer is stable.
      "stretched"  the reference blob scaled along x to 0.62 of the cell's a edge
      "two_lobes"  the blob shrunk to half its size, every third atom moved 0.42 a along x
    then noise, a rigid translation to anywhere in the cell, and the wrap into the cell (the group arrives PBC-broken)."""
    rng = np.random.default_rng(seed)
    p = np.asarray(ref_pos, np.float64)
    c = p.mean(axis=0)
    d = p - c
    ax = float(box9[0])
    if kind == "stretched":
        d[:, 0] *= 0.62 * ax / (d[:, 0].max() - d[:, 0].min())
    elif kind == "two_lobes":
        d *= 0.5
        d[::3, 0] += 0.42 * ax
    else:
        raise ValueError(kind)
    frac = d @ np.linalg.inv(box_matrix(box9))
    assert (frac.max(axis=0) - frac.min(axis=0)).max() > 0.52, "the group does not span half the cell"
    d += rng.normal(0.0, noise, d.shape)
    t = rng.uniform(0.0, 1.0, 3) @ box_matrix(box9)
    return wrap_into_cell(c + d + t, box9)
