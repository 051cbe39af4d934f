#!/usr/bin/env python3
"""BASELINE configs[4] at length, on the record: a 5e5-atom truncated-octahedron trajectory (water-like, written with the library's
encoder at precision 1000), >= 1000 frames through the REAL decoder, per-frame COM of a 30 000-atom solute + centring on it + wrap
of all atoms (group_get_com, atoms_center_mass), double-buffered.  Two ingest designs, each with its stages timed ALONE and then
together; overlap_efficiency = max(stage alone) / wall of the pipeline (SURVEY 8(d) "end-to-end ... overlap efficiency"):

  host-decode   T threads decode whole frames into pinned buffers (one frame per thread) || hipMemcpyAsync H2D on the copy stream ||
                the analyses on the compute stream -- north_star's "pinned hipMemcpyAsync double-buffer that overlaps xtc/trr
                decode on the host with kernel execution", literally
  device-unpack the host only SKIMS the framing, the compressed stream crosses PCIe, k_xtc_unpack decodes batches of B frames on
                the copy stream (gr_xtc_read_frames_device) || the batched analyses of the previous batch

    python tools/c5_pipeline.py [--atoms 500000] [--frames 1024] [--threads 16] [--batch 64]
"""
import argparse
import json
import os
import queue
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
from groan_rs_amd import workload as W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--atoms", type=int, default=500_000)
    ap.add_argument("--frames", type=int, default=1024, help="frames that go through each pipeline (the file holds --file-frames of them, read round and round)")
    ap.add_argument("--file-frames", type=int, default=128)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    n, NF, FF, T, B = a.atoms - a.atoms % 3, a.frames, a.file_frames, a.threads, a.batch
    box = W.c5_box(17.5)
    rng = np.random.default_rng(5)
    nm = n // 3
    frac = rng.random((nm, 3))
    ctr = frac @ W.box_matrix(box)
    base = np.repeat(ctr, 3, axis=0)
    base[1::3] += rng.normal(0, 0.055, (nm, 3)); base[2::3] += rng.normal(0, 0.055, (nm, 3))     # O H H within ~0.1 nm
    masses = np.array([15.999, 1.008, 1.008], np.float32)[np.arange(n) % 3]
    n_slots = max(2 * B, 16)
    s = G.System(n, masses=masses, n_slots=n_slots)
    s.group_create_from_ranges("Solute", [(0, 29_999)])
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "water_octa.xtc")
    with G.XtcWriter(path) as w:
        for k in range(8):
            s.set_frame((base + rng.normal(0, 0.02, base.shape)).astype(np.float32), box, slot=k)
        for f0 in range(0, FF, 8):
            w.write_slots(s, 0, 8, precision=1000.0, host_threads=T)
    x = G.XtcFile(path)
    assert x.n_atoms == n and x.n_frames == FF
    fbytes = os.path.getsize(path) / FF
    out = {"config": "BASELINE configs[4]: truncated octahedron scaled to %d atoms, xtc decode || H2D || COM + centre/wrap, double-buffered, 1 GPU" % n,
           "n_atoms": n, "frames_through_each_pipeline": NF, "frames_in_file": FF, "host_threads": T, "compressed_MB_per_frame": round(fbytes / 1e6, 3),
           "raw_MB_per_frame": round(12.0 * n / 1e6, 3)}

    # ------------------------------------------------------------------ design 1: host decode || H2D || analyses
    n_buf = 2 * T
    staging = [G.pinned_array((n, 3)) for _ in range(n_buf)]

    def host_decode_alone(frames):
        nxt = [0]; lock = threading.Lock()
        def dec(b):
            while True:
                with lock:
                    f = nxt[0]; nxt[0] += 1
                if f >= frames:
                    return
                x.read_frame(f % FF, out=staging[b][0])
        t0 = time.perf_counter()
        th = [threading.Thread(target=dec, args=(b,)) for b in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
        return time.perf_counter() - t0

    host_decode_alone(2 * T)                                                         # (page in the file, warm the threads)
    nd = min(NF, 256)
    t_dec = host_decode_alone(nd) / nd
    x.read_frame(0, out=staging[0][0]); bx = x.read_frame(0)[1]
    for k in range(4):
        s.upload_async(staging[0][0], bx, k % n_slots); s.upload_wait(k % n_slots)
    t0 = time.perf_counter()
    for k in range(64):                                                               # H2D alone (pinned source, copy stream, the tile kernel behind it)
        s.upload_async(staging[k % n_buf][0], bx, k % n_slots)
    for k in range(n_slots):
        s.upload_wait(k)
    s.sync()
    t_h2d = (time.perf_counter() - t0) / 64
    for _ in range(2):
        s.group_get_com("Solute", slot=0); s.atoms_center_mass("Solute", G.Dimension.XYZ, slot=0)
    s.sync()
    t0 = time.perf_counter()
    for k in range(64):                                                               # the per-frame analyses alone (each call ends with its own read-back)
        s.group_get_com("Solute", slot=k % n_slots); s.atoms_center_mass("Solute", G.Dimension.XYZ, slot=k % n_slots)
    s.sync()
    t_ana = (time.perf_counter() - t0) / 64

    free_q, ready = queue.Queue(), {}
    cond = threading.Condition()
    for b in range(n_buf):
        free_q.put(b)
    next_frame = [0]
    lock = threading.Lock()

    def decoder():
        while True:
            with lock:
                f = next_frame[0]
                if f >= NF:
                    return
                next_frame[0] += 1
            b = free_q.get()
            _, fbox, _, _, _ = x.read_frame(f % FF, out=staging[b][0])
            with cond:
                ready[f] = (b, fbox)
                cond.notify_all()

    ths = [threading.Thread(target=decoder) for _ in range(T)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    com_sum = np.zeros(3)
    in_flight = []                                                                    # (staging buffer, slot) whose copies have not been waited for
    for f in range(NF):
        with cond:
            while f not in ready:
                cond.wait()
            b, fbox = ready.pop(f)
        slot = f % n_slots
        s.upload_async(staging[b][0], fbox, slot)
        in_flight.append((b, slot))
        if len(in_flight) > 2:
            pb, pslot = in_flight.pop(0)
            s.upload_wait(pslot); free_q.put(pb)
        com_sum += s.group_get_com("Solute", slot=slot)
        s.atoms_center_mass("Solute", G.Dimension.XYZ, slot=slot)
    s.sync()
    t_pipe = time.perf_counter() - t0
    [t.join() for t in ths]
    stage = {"host_decode_%d_threads" % T: t_dec, "h2d_copy_stream": t_h2d, "analyses_per_frame_calls": t_ana}
    out["host_decode_pipeline"] = {
        "stage_seconds_per_frame_alone": {k: round(v, 7) for k, v in stage.items()},
        "stage_frames_per_s_alone": {k: round(1.0 / v, 1) for k, v in stage.items()},
        "pipeline_frames_per_s": round(NF / t_pipe, 1), "pipeline_wall_s": round(t_pipe, 3),
        "overlap_efficiency": round(max(stage.values()) * NF / t_pipe, 3), "bound_by": max(stage, key=stage.get),
        "h2d_GBps": round(12.0 * n / t_h2d / 1e9, 1)}
    for _, ptr in staging:
        G.pinned_free(ptr)

    # ------------------------------------------------------------------ design 2: host skim || H2D of the compressed stream || k_xtc_unpack || batched analyses
    for wv in range(2):
        x.read_frames_device(s, 0, B, first_slot=wv * B, host_threads=T)
    s.sync()
    nb = NF // B
    t0 = time.perf_counter()
    for k in range(nb):                                                               # ingest alone (its own three stages overlap inside the call)
        x.read_frames_device(s, (k * B) % FF, B, first_slot=(k % 2) * B, host_threads=T)
    for k in range(2 * B):
        s.upload_wait(k)
    s.sync()
    t_ing = (time.perf_counter() - t0) / (nb * B)
    for _ in range(2):
        s.group_get_com_batch("Solute", 0, B); s.atoms_center_batch("Solute", 0, B, G.Dimension.XYZ, weighted=True)
    s.sync()
    t0 = time.perf_counter()
    for k in range(8):                                                                # batched analyses alone
        s.group_get_com_batch("Solute", (k % 2) * B, B); s.atoms_center_batch("Solute", (k % 2) * B, B, G.Dimension.XYZ, weighted=True)
    s.sync()
    t_anab = (time.perf_counter() - t0) / (8 * B)
    t0 = time.perf_counter()
    x.read_frames_device(s, 0, B, first_slot=0, host_threads=T)
    for k in range(nb):
        if k + 1 < nb:
            x.read_frames_device(s, ((k + 1) * B) % FF, B, first_slot=((k + 1) % 2) * B, host_threads=T)
        cb, _ = s.group_get_com_batch("Solute", (k % 2) * B, B)
        s.atoms_center_batch("Solute", (k % 2) * B, B, G.Dimension.XYZ, weighted=True)
    s.sync()
    t_pipe2 = time.perf_counter() - t0
    stage2 = {"ingest_skim_h2d_unpack": t_ing, "analyses_batched": t_anab}
    out["device_unpack_pipeline"] = {
        "batch": B, "stage_seconds_per_frame_alone": {k: round(v, 7) for k, v in stage2.items()},
        "stage_frames_per_s_alone": {k: round(1.0 / v, 1) for k, v in stage2.items()},
        "pipeline_frames_per_s": round(nb * B / t_pipe2, 1), "pipeline_wall_s": round(t_pipe2, 3),
        "overlap_efficiency": round(max(stage2.values()) * nb * B / t_pipe2, 3), "bound_by": max(stage2, key=stage2.get),
        "pcie_GBps_compressed": round(fbytes / t_ing / 1e9, 1)}
    print(json.dumps(out, indent=1))
    x.close(); s.close(); os.remove(path)


if __name__ == "__main__":
    main()
