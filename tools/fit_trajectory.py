#!/usr/bin/env python3
"""End-to-end use of the library on a trajectory file: RMSD-fit every frame of an xtc to its first frame and write the
fitted trajectory -- the reference's `system.xtc_iter(f)?.convert_and_analyze(RMSDConverterAnalyzer)` + `XtcWriter` loop
(src/system/rmsd.rs:170-251, src/io/xtc_io/mod.rs:256-331) with every stage on the fast path:

    xtc bytes --(host: skim framing)--> device unpack --> RMSD + fit kernels --> D2H --> host encode --> fitted xtc

    python tools/fit_trajectory.py in.xtc out.xtc [--group FIRST:LAST] [--batch 64] [--masses masses.npy]

Prints the per-frame RMSD and the throughput.  Masses default to 1 (the RMSD is then unweighted)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import groan_rs_amd as G


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("input"); ap.add_argument("output")
    ap.add_argument("--group", default=None, help="atom range FIRST:LAST (inclusive, 0-based) to fit on; default all atoms")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--masses", default=None)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    x = G.XtcFile(a.input)
    n, nf = x.n_atoms, x.n_frames
    masses = np.load(a.masses).astype(np.float32) if a.masses else np.ones(n, np.float32)
    B = min(a.batch, nf)
    cur = G.System(n, masses=masses, n_slots=B)
    pos0, box0, _, _, prec = x.read_frame(0)
    ref = G.System(n, masses=masses, box=box0, positions=pos0)
    group = "all"
    if a.group:
        lo, hi = (int(v) for v in a.group.split(":"))
        for s in (ref, cur):
            s.group_create_from_ranges("fit", [(lo, hi)])
        group = "fit"
    plan = G.RMSDPlan(ref, cur, group)
    rmsd = np.zeros(nf, np.float32)
    t0 = time.perf_counter()
    with G.XtcWriter(a.output) as w:
        for f0 in range(0, nf, B):
            nb = min(B, nf - f0)
            steps, times = x.read_frames_device(cur, f0, nb, host_threads=a.threads)
            r, st = plan.rmsd_fit(0, nb)
            rmsd[f0:f0 + nb] = r
            w.write_slots(cur, 0, nb, steps=steps.astype(np.int64), times=times, precision=prec if prec > 0 else 1000.0, host_threads=a.threads)
    dt = time.perf_counter() - t0
    for f in range(nf):
        print("%6d %10.6f" % (f, rmsd[f]))
    print("# %d frames x %d atoms in %.3f s = %.1f frames/s (decode on device, fit on device, encode on %s host threads)"
          % (nf, n, dt, nf / dt, a.threads or "up to 16"), file=sys.stderr)


if __name__ == "__main__":
    main()
