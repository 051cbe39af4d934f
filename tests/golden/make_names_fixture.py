#!/usr/bin/env python3
"""Residue numbers / residue names / atom names / atom numbers of the reference's test_files/example.gro as a compact fixture
for the selection-language tests (tests/test_select.py): example_names.npz.  Data only (the columns of a data file the
reference's own tests read, src/system/groups.rs tests); run in the build container where /root/reference is mounted."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TF = "/root/reference/test_files"


def main():
    lines = open(os.path.join(TF, "example.gro")).read().split("\n")
    n = int(lines[1])
    resid = np.zeros(n, np.uint32); atomid = np.zeros(n, np.uint32)
    resname, atomname = [], []
    for i in range(n):
        ln = lines[2 + i]
        resid[i] = int(ln[0:5]); resname.append(ln[5:10].strip()); atomname.append(ln[10:15].strip()); atomid[i] = int(ln[15:20])
    np.savez_compressed(os.path.join(HERE, "example_names.npz"), resid=resid, atomid=atomid,
                        resname=np.array(resname, "S5"), atomname=np.array(atomname, "S5"))
    print(n, "atoms;", len(set(resname)), "residue names;", os.path.getsize(os.path.join(HERE, "example_names.npz")), "bytes")


if __name__ == "__main__":
    main()
