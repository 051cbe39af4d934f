#!/usr/bin/env python3
"""Workload for tools/pmc_center.sh: a few calls of atoms_center_mass(all) (one pass, dodecahedron) and atoms_translate / atoms_center(tenth) in an
orthorhombic cell (k_translate_wrap_rows), 1e6 atoms, 64 frames per call."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import groan_rs_amd as G
from groan_rs_amd import workload as W
n, NF = 1_000_000, 64
for box in (W.c4_box(), W.box_from_lengths_angles([24.0, 23.0, 22.0], [90.0, 90.0, 90.0])):
    s = G.System(n, masses=W.masses_cycle(n), n_slots=NF + 1)
    s.synth_reference(NF, box, 0.2 * float(min(box[:3])), 1)
    s.synth_frames(NF, 0, NF, 0, 0.05, 1)
    for _ in range(3):
        s.atoms_center_batch("all", 0, NF, weighted=True)
        s.group_translate_batch(None, [0.3, -0.2, 0.1], 0, NF)
    s.close()
