#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the reference's DATA files.

Run in the build container only (needs /root/reference and oracle/_ref/libxdrfile_ref.so, the
reference's vendored C xdrfile compiled where it lies by `make -C oracle ref`):

    python tests/golden/make_golden.py

What is written is data only -- decoded coordinates, boxes, masses, index ranges -- never any
reference source text.  Known-answer VALUES pinned by the reference's own unit tests live in
tests/test_oracle_golden.py next to the file:line they come from.

Outputs (numpy .npz, float32 unless noted):
  example.npz        example.gro positions (16844x3), box9, masses of the 61 Protein beads read
                     from example.tpr, ndx groups Protein/Membrane/... as inclusive block ranges
  short_traj.npz     short_trajectory.xtc: 11 frames of the Protein(61) + 200 spread atoms,
                     boxes, and the same atoms decoded from short_trajectory_fit.xtc and
                     short_trajectory_broken_fit.xtc (the reference's RMSD-fit goldens)
  aa_peptide.npz     aa_membrane_peptide.gro positions (32817x3), box, element masses
                     (first-letter rule of src/config/elements.yaml), @protein / @membrane blocks,
                     and the 363 peptide atoms of the 21 frames of aa_membrane_peptide.xtc
  tric_small.npz     the 50-atom triclinic / octahedron / dodecahedron frames (xtc) + boxes
"""
import ctypes as C
import os
import re
import sys

import numpy as np

REF = os.environ.get("GROAN_REFERENCE", "/root/reference")
TF = os.path.join(REF, "test_files")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def xdr():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libxdrfile_ref.so"))
    lib.xdrfile_open.restype = C.c_void_p
    lib.xdrfile_open.argtypes = [C.c_char_p, C.c_char_p]
    lib.xdrfile_close.argtypes = [C.c_void_p]
    lib.read_xtc_natoms.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.read_xtc.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float),
                             C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    return lib


def read_xtc(path):
    """-> (coords [F,N,3] f32, boxes [F,3,3] f32 row-major box vectors, steps, times, precision)"""
    lib = xdr()
    n = C.c_int(0)
    assert lib.read_xtc_natoms(path.encode(), C.byref(n)) == 0
    n = n.value
    fh = lib.xdrfile_open(path.encode(), b"r")
    assert fh
    frames, boxes, steps, times = [], [], [], []
    prec = C.c_float(0)
    while True:
        x = np.zeros((n, 3), np.float32)
        box = np.zeros((3, 3), np.float32)
        step = C.c_int(0)
        time = C.c_float(0)
        rc = lib.read_xtc(fh, n, C.byref(step), C.byref(time), box.ctypes.data, x.ctypes.data, C.byref(prec))
        if rc != 0:
            break
        frames.append(x); boxes.append(box); steps.append(step.value); times.append(time.value)
    lib.xdrfile_close(fh)
    return np.stack(frames), np.stack(boxes), np.array(steps), np.array(times, np.float32), prec.value


def matrix2box9(m):
    """xtc 3x3 (rows = box vectors) -> gro-order box9 (src/io/xdrfile.rs:170-187)"""
    return np.array([m[0][0], m[1][1], m[2][2], m[0][1], m[0][2], m[1][0], m[1][2], m[2][0], m[2][1]], np.float32)


def read_gro(path):
    with open(path) as fh:
        lines = fh.read().split("\n")
    n = int(lines[1])
    pos = np.zeros((n, 3), np.float32)
    resname, atomname = [], []
    for i in range(n):
        ln = lines[2 + i]
        resname.append(ln[5:10].strip())
        atomname.append(ln[10:15].strip())
        pos[i] = [np.float32(ln[20:28]), np.float32(ln[28:36]), np.float32(ln[36:44])]
    b = [np.float32(v) for v in lines[2 + n].split()]
    box9 = np.zeros(9, np.float32)
    box9[: len(b)] = b
    return pos, resname, atomname, box9


def read_ndx(path):
    groups, cur = {}, None
    with open(path) as fh:
        for ln in fh:
            m = re.match(r"\s*\[\s*(.+?)\s*\]", ln)
            if m:
                cur = m.group(1)
                groups[cur] = []
            elif cur is not None:
                groups[cur].extend(int(t) - 1 for t in ln.split())
    return groups


def to_blocks(indices):
    """sorted unique indices -> inclusive [start,end] blocks"""
    idx = np.unique(np.asarray(indices, np.int64))
    if idx.size == 0:
        return np.zeros((0, 2), np.uint64)
    brk = np.where(np.diff(idx) != 1)[0]
    starts = np.concatenate([[idx[0]], idx[brk + 1]])
    ends = np.concatenate([idx[brk], [idx[-1]]])
    return np.stack([starts, ends], 1).astype(np.uint64)


def main():
    # ---------------- example.gro / example.tpr / index.ndx ----------------
    pos, resname, atomname, box9 = read_gro(os.path.join(TF, "example.gro"))
    ndx = read_ndx(os.path.join(TF, "index.ndx"))
    raw = open(os.path.join(TF, "example.tpr"), "rb").read()
    # Protein moleculetype atom records in example.tpr: big-endian f32 mass at byte 16834, stride 32
    m61 = np.array([np.frombuffer(raw[16834 + 32 * i: 16838 + 32 * i], ">f4")[0] for i in range(61)], np.float32)
    assert set(np.unique(m61)) <= {36.0, 54.0, 72.0} and m61.sum() == 3204.0
    # positions stored in the tpr are the gro's (checked: identical to 0.0)
    tpr_pos = np.frombuffer(raw[43676: 43676 + 12 * pos.shape[0]], ">f4").reshape(-1, 3).astype(np.float32)
    assert np.array_equal(tpr_pos, pos)
    out = {"pos": pos, "box9": box9, "protein_masses": m61}
    for g in ("Protein", "Membrane", "W", "ION", "Transmembrane", "Transmembrane_all", "Backbone", "SideChain"):
        out["blocks_" + g] = to_blocks(ndx[g])
    np.savez_compressed(os.path.join(HERE, "example.npz"), **out)
    prot = np.arange(61)
    assert np.array_equal(np.unique(ndx["Protein"]), prot)

    # ---------------- short_trajectory + fit goldens ----------------
    n = pos.shape[0]
    extra = np.linspace(61, n - 1, 200).astype(np.int64)
    keep = np.concatenate([prot, extra])
    X, B, steps, times, prec = read_xtc(os.path.join(TF, "short_trajectory.xtc"))
    Xf, Bf, _, _, _ = read_xtc(os.path.join(TF, "short_trajectory_fit.xtc"))
    Xb, Bb, _, _, _ = read_xtc(os.path.join(TF, "short_trajectory_broken_fit.xtc"))
    assert X.shape == Xf.shape == Xb.shape == (11, n, 3)
    np.savez_compressed(
        os.path.join(HERE, "short_traj.npz"),
        keep=keep.astype(np.uint64), frames=X[:, keep], boxes9=np.stack([matrix2box9(b) for b in B]),
        fit=Xf[:, keep], broken_fit=Xb[:, keep], steps=steps, times=times, precision=np.float32(prec),
        gro_keep=pos[keep],
    )

    # ---------------- aa_membrane_peptide ----------------
    apos, ares, aname, abox9 = read_gro(os.path.join(TF, "aa_membrane_peptide.gro"))
    # element masses: the queries of src/config/elements.yaml that match this system
    # (hydrogen r'^[1-9]?[Hh].*', carbon/nitrogen/oxygen/phosphorus first letter, ions by name)
    masses = np.zeros(len(aname), np.float32)
    for i, (nm, rn) in enumerate(zip(aname, ares)):
        ion = rn in ("NA", "CL") or nm in ("NA", "CL")
        if re.match(r"^[1-9]?[Hh].*", nm): masses[i] = 1.0079
        elif ion and nm.upper().startswith("NA"): masses[i] = 22.98970
        elif ion and nm.upper().startswith("CL"): masses[i] = 35.45300
        elif nm[0] in "Cc": masses[i] = 12.0107
        elif nm[0] in "Nn": masses[i] = 14.00670
        elif nm[0] in "Oo": masses[i] = 15.99940
        elif nm[0] in "Pp": masses[i] = 30.97380
        elif nm[0] in "Ss": masses[i] = 32.06500
        else: raise SystemExit("unassigned element for %s/%s" % (rn, nm))
    protein_res = {"LEU", "SER", "LYS"}
    pep = [i for i, r in enumerate(ares) if r in protein_res]
    mem = [i for i, r in enumerate(ares) if re.match(r"^[A-Za-z]{2}(PA|PC|PE|PG|PS|PI|GL|DG)$", r)]
    pep_blocks, mem_blocks = to_blocks(pep), to_blocks(mem)
    assert pep_blocks.tolist() == [[0, 362]]
    AX, AB, asteps, atimes, aprec = read_xtc(os.path.join(TF, "aa_membrane_peptide.xtc"))
    assert AX.shape == (21, apos.shape[0], 3)
    np.savez_compressed(
        os.path.join(HERE, "aa_peptide.npz"),
        pos=apos, box9=abox9, masses=masses, blocks_peptide=pep_blocks, blocks_membrane=mem_blocks,
        traj_peptide=AX[:, :363], traj_boxes9=np.stack([matrix2box9(b) for b in AB]),
    )

    # ---------------- small non-orthogonal boxes ----------------
    tri = {}
    for name in ("triclinic", "octahedron", "dodecahedron"):
        TX, TB, _, _, _ = read_xtc(os.path.join(TF, name + "_trajectory.xtc"))
        gp, _, _, gb = read_gro(os.path.join(TF, name + ".gro"))
        tri[name + "_frames"] = TX
        tri[name + "_boxes9"] = np.stack([matrix2box9(b) for b in TB])
        tri[name + "_gro_pos"] = gp
        tri[name + "_gro_box9"] = gb
    np.savez_compressed(os.path.join(HERE, "tric_small.npz"), **tri)
    # ---------------- xtc data files for the decoder tests (reference data files, not source) ----------------
    import hashlib
    import shutil
    sums = {}
    for name in ("triclinic_trajectory.xtc", "octahedron_trajectory.xtc", "dodecahedron_trajectory.xtc", "short_trajectory.xtc"):
        shutil.copyfile(os.path.join(TF, name), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)
        FX, FB, fs, ft, fp = read_xtc(os.path.join(TF, name))
        sums[name] = {"sha256_coords_f32le": hashlib.sha256(FX.astype("<f4").tobytes()).hexdigest(), "n_frames": int(FX.shape[0]),
                      "n_atoms": int(FX.shape[1]), "steps": [int(v) for v in fs], "times": [float(v) for v in ft], "precision": float(fp),
                      "boxes_rowmajor": [[float(v) for v in b.ravel()] for b in FB]}
    import json
    json.dump(sums, open(os.path.join(HERE, "xtc_expected.json"), "w"), indent=1)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz") or f.endswith(".xtc"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
