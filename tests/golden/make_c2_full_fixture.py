#!/usr/bin/env python3
"""BASELINE configs[1] at its real shape: the 32 817-atom system of aa_membrane_peptide, not the 363 peptide atoms
on their own.  Fixture aa_full.npz = frames 0 and 20 of the reference's test_files/aa_membrane_peptide.xtc decoded with the reference's
own vendored xdrfile (oracle/_ref, as make_golden.py does) + their boxes, and the residue numbers / residue names / atom names / atom
numbers of aa_membrane_peptide.gro (what the selection language needs for `@protein`).  Data only.  Run in the build container:

    make -C oracle all && python tests/golden/make_c2_full_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG   # noqa: E402


def main():
    tf = MG.TF
    X, B, steps, times, prec = MG.read_xtc(os.path.join(tf, "aa_membrane_peptide.xtc"))
    assert X.shape == (21, 32817, 3)
    lines = open(os.path.join(tf, "aa_membrane_peptide.gro")).read().split("\n")
    n = int(lines[1])
    resid = np.zeros(n, np.uint32); atomid = np.zeros(n, np.uint32); resname, atomname = [], []
    for i in range(n):
        ln = lines[2 + i]
        resid[i] = int(ln[0:5]); resname.append(ln[5:10].strip()); atomname.append(ln[10:15].strip()); atomid[i] = int(ln[15:20])
    keep = [0, 20]
    out = os.path.join(HERE, "aa_full.npz")
    np.savez_compressed(out, frames=X[keep], boxes9=np.stack([MG.matrix2box9(B[k]) for k in keep]), frame_index=np.array(keep), steps=np.asarray(steps)[keep],
                        precision=np.float32(prec), resid=resid, atomid=atomid, resname=np.array(resname, "S5"), atomname=np.array(atomname, "S5"))
    print(n, "atoms,", len(keep), "frames,", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
