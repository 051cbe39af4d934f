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
    cur.set_persistent(2)
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
    laps = t[:, : W - 1, 9:15].copy()                         # stamps 9.. are durations of wave 15's phases, not times
    t = t[:, :, :9]
    t0 = t[t > 0].min()
    t = np.where(t > 0, t - t0, np.nan)
    total = np.nanmax(t)
    print("frames %d  workgroups %d (last = finalizers)  total %.1f us  -> %.2f us/frame" % (F, W, total, total / F))
    def stat(name, x):
        x = x[np.isfinite(x)]
        print("  %-44s mean %7.2f  p50 %7.2f  p95 %7.2f  max %7.2f us" % (name, x.mean(), np.median(x), np.percentile(x, 95), x.max()))
    c = t[:, : W - 1, :]                                      # compute workgroups (comm wave stamps)
    fz = t[:, W - 1, :]                                       # finalizer stamps
    for k, name in enumerate(("tile landed in LDS", "sums of 4 atoms/lane", "wave reduce + LDS arrive", "wait for LDS ready", "transform + rmsd terms", "store + rmsd wave sum")):
        stat("wave 15: " + name, laps[:, :, k])
    stat("wave 15: all phases", laps.sum(axis=2))
    stat("comm: A start -> all waves arrived (0->1)", c[:, :, 1] - c[:, :, 0])
    stat("comm: publish record + arrive (1->2)", c[:, :, 2] - c[:, :, 1])
    stat("comm: wait for ready (3->4)", c[:, :, 4] - c[:, :, 3])
    stat("comm: own C phase (4->5)", c[:, :, 5] - c[:, :, 4])
    stat("comm: whole iteration (0 -> next 0)", np.diff(c[:, :, 0], axis=0))
    last = np.nanmax(c[:, :, 2], axis=1); first = np.nanmin(c[:, :, 2], axis=1)
    stat("arrival skew per frame (first->last)", (last - first)[:, None])
    stat("last arrival -> finalizer sees it", (fz[:, 2] - last)[:, None])
    stat("finalizer: sum records (2->6)", (fz[:, 6] - fz[:, 2])[:, None])
    stat("finalizer: math (6->7)", (fz[:, 7] - fz[:, 6])[:, None])
    stat("finalizer: publish (7->8)", (fz[:, 8] - fz[:, 7])[:, None])
    stat("published -> first wg sees ready", (np.nanmin(c[:, :, 4], axis=1) - fz[:, 8])[:, None])
    stat("published -> last wg sees ready", (np.nanmax(c[:, :, 4], axis=1) - fz[:, 8])[:, None])
    stat("last arrival -> published (latency L)", (fz[:, 8] - last)[:, None])
    stat("frame period (publish to publish)", np.diff(fz[:, 8])[:, None])


if __name__ == "__main__":
    main()
