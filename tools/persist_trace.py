#!/usr/bin/env python3
"""Run one persistent RMSD-fit batch with GR_PS_TRACE and print where a frame's time goes.

    GR_PS_TRACE=/tmp/ps_trace.bin python tools/persist_trace.py [n_atoms] [n_frames]

Stamps are s_memrealtime ticks (100 MHz): per (frame, workgroup) 0 A start, 1 sums done, 2 arrived, 3 C wait start,
4 ready seen, 5 C done; finalizer only: 6 records summed, 7 math done, 8 published."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    path = os.environ.setdefault("GR_PS_TRACE", "/tmp/ps_trace.bin")
    import groan_rs_amd as G
    import oracle_lib as O
    box = O.box_from_lengths_angles([24.18, 24.18, 24.18], [60.0, 60.0, 90.0])
    masses = np.array([1.008, 12.011, 14.007, 15.999], np.float32)[np.arange(n) % 4]
    cur = G.System(n, masses=masses, n_slots=nf + 1)
    cur.set_persistent(True)
    cur.synth_reference(nf, box, 0.2 * float(min(box[:3])), 1)
    ref = G.System(n, masses=masses, box=box, positions=cur.get_positions(nf))
    plan = G.RMSDPlan(ref, cur, "all")
    for rep in range(3):
        cur.synth_frames(nf, 0, nf, 0, 0.05, 1)
        r, st = plan.rmsd_fit(0, nf)
    assert plan.last_persistent() and (st == 0).all()
    raw = np.fromfile(path, dtype=np.uint64)
    F, W, K = int(raw[0]), int(raw[1]), int(raw[2])
    t = raw[4:].reshape(F, W, K).astype(np.float64) * 0.01    # us
    t0 = t[t > 0].min()
    t = np.where(t > 0, t - t0, np.nan)
    total = np.nanmax(t)
    print("frames %d  workgroups %d  total %.1f us  -> %.2f us/frame" % (F, W, total, total / F))
    def stat(name, x):
        x = x[np.isfinite(x)]
        print("  %-38s mean %7.2f  p50 %7.2f  p95 %7.2f  max %7.2f us" % (name, x.mean(), np.median(x), np.percentile(x, 95), x.max()))
    stat("A: load + sums (0->1)", t[:, :, 1] - t[:, :, 0])
    stat("A: wg reduce + publish + arrive (1->2)", t[:, :, 2] - t[:, :, 1])
    stat("C: wait for ready (3->4)", t[:, :, 4] - t[:, :, 3])
    stat("C: transform + store (4->5)", t[:, :, 5] - t[:, :, 4])
    last = np.nanmax(t[:, :, 2], axis=1)
    first = np.nanmin(t[:, :, 2], axis=1)
    stat("arrival skew per frame (first->last)", (last - first)[:, None])
    fin6 = np.nanmax(t[:, :, 6], axis=1); fin7 = np.nanmax(t[:, :, 7], axis=1); fin8 = np.nanmax(t[:, :, 8], axis=1)
    stat("finalizer: sum 256 records (2->6)", (fin6 - last)[:, None])
    stat("finalizer: math (6->7)", (fin7 - fin6)[:, None])
    stat("finalizer: publish (7->8)", (fin8 - fin7)[:, None])
    seen = np.nanmin(t[:, :, 4], axis=1)
    stat("published -> first wg sees ready", (seen - fin8)[:, None])
    stat("published -> last wg sees ready", (np.nanmax(t[:, :, 4], axis=1) - fin8)[:, None])
    per = np.diff(fin8)
    stat("frame period (publish to publish)", per[:, None])


if __name__ == "__main__":
    main()
