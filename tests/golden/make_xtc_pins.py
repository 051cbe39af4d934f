#!/usr/bin/env python3
"""Writes tests/golden/xtc_pins.json: the sha256 of every synthetic test trajectory (tests/xtc_cases.py) as encoded by the
REFERENCE's own xtc writer -- its vendored C xdrfile, compiled where it lies by oracle/Makefile into oracle/_ref (build
container only).  The GPU suite writes the same cases with the library's own encoder and checks them against these pins.

    python tests/golden/make_xtc_pins.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import xtc_cases as XC                      # noqa: E402
from test_xtc_decoder import write_with_ref   # noqa: E402  (ctypes binding of oracle/_ref/libxdrfile_ref.so)


def main():
    pins = {}
    with tempfile.TemporaryDirectory() as d:
        for case in XC.CASES:
            frames, box, prec = XC.branch_case(case)
            p = os.path.join(d, case + ".xtc")
            write_with_ref(p, frames, box, prec)
            pins[case] = {"sha256": XC.sha256_file(p), "bytes": os.path.getsize(p), "n_atoms": int(frames[0].shape[0]), "n_frames": len(frames), "precision": prec}
        frames, box9, boxm = XC.octahedron_case()
        p = os.path.join(d, "octa.xtc")
        write_with_ref(p, [frames[f % 4] for f in range(32)], boxm, 1000.0)
        pins["octahedron_5e5_x32"] = {"sha256": XC.sha256_file(p), "bytes": os.path.getsize(p), "n_atoms": 500000, "n_frames": 32, "precision": 1000.0}
    json.dump(pins, open(XC.PINS, "w"), indent=1, sort_keys=True)
    for k, v in pins.items():
        print(k, v["bytes"], v["sha256"][:16])


if __name__ == "__main__":
    main()
