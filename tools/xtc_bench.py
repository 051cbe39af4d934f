#!/usr/bin/env python3
"""config 5 stage rates: xtc decode of a water-like 5e5-atom truncated-octahedron trajectory.
Writes NF frames with the library's encoder, then times (frames/s): the host decoder on T threads, gr_xtc_read_frames_device
(host skim + H2D of the compressed stream + k_xtc_unpack) for several batch sizes, and the same followed by the batched
COM + centre/wrap analyses."""
import argparse
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import groan_rs_amd as G
import oracle_lib as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--atoms", type=int, default=500_000)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--passes", type=int, default=4, help="passes over the file in the device-unpack loops")
    a = ap.parse_args()
    n, NF, T = a.atoms - a.atoms % 3, a.frames, a.threads
    PASSES = a.passes
    box = O.box_from_lengths_angles([17.5] * 3, [70.53, 109.47, 70.53])
    rng = np.random.default_rng(5)
    nm = n // 3
    frac = rng.random((nm, 3))
    ctr = frac[:, :1] * [box[0], 0, 0] + frac[:, 1:2] * [box[5], box[1], 0] + frac[:, 2:] * [box[7], box[8], box[2]]
    base = np.repeat(ctr, 3, axis=0)
    base[1::3] += rng.normal(0, 0.055, (nm, 3)); base[2::3] += rng.normal(0, 0.055, (nm, 3))     # O H H within ~0.1 nm
    masses = np.array([15.999, 1.008, 1.008], np.float32)[np.arange(n) % 3]
    slots = 256
    s = G.System(n, masses=masses, n_slots=slots)
    s.group_create_from_ranges("Solute", [(0, 29_999)])
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "water_octa.xtc")
    t0 = time.perf_counter()
    with G.XtcWriter(path) as w:
        for k in range(8):
            s.set_frame((base + rng.normal(0, 0.02, base.shape)).astype(np.float32), box, slot=k)
        for f0 in range(0, NF, 8):
            w.write_slots(s, 0, 8, precision=1000.0, host_threads=T)       # the same eight frames over and over (content is irrelevant to the rates)
    t_write = time.perf_counter() - t0
    x = G.XtcFile(path)
    assert x.n_atoms == n and x.n_frames == NF
    out = {"n_atoms": n, "n_frames": NF, "host_threads": T, "passes": PASSES, "file_MB": round(os.path.getsize(path) / 1e6, 1),
           "compressed_bytes_per_atom": round(os.path.getsize(path) / NF / n, 3), "encode_frames_per_s": round(NF / t_write, 1)}
    # host decoder, T threads
    bufs = [np.zeros((n, 3), np.float32) for _ in range(T)]
    nxt = [0]; lock = threading.Lock()
    def dec(b):
        while True:
            with lock:
                f = nxt[0]; nxt[0] += 1
            if f >= NF:
                return
            x.read_frame(f, out=bufs[b])
    t0 = time.perf_counter()
    th = [threading.Thread(target=dec, args=(b,)) for b in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    out["host_decode_frames_per_s"] = round(NF / (time.perf_counter() - t0), 1)
    for B in (8, 16, 32, 64, 128):
        if 2 * B > slots or B > NF:
            continue
        for w in range(2):                                   # both staging banks, both halves of the slots
            x.read_frames_device(s, 0, B, first_slot=w * B, host_threads=T)
        s.sync()
        t0 = time.perf_counter()
        for k in range(PASSES * NF // B):                    # several passes over the file: the steady-state rate, not the drain of a 4-call run
            x.read_frames_device(s, (k * B) % NF, B, first_slot=(k % 2) * B, host_threads=T)
        for k in range(2 * B):
            s.upload_wait(k)
        out["device_unpack_batch_%d_frames_per_s" % B] = round(PASSES * NF / (time.perf_counter() - t0), 1)
        # decode of batch k + 1 (host skim, H2D, unpack on the copy stream) beside the analyses of batch k (compute stream)
        t0 = time.perf_counter()
        x.read_frames_device(s, 0, B, first_slot=0, host_threads=T)
        for k, f0 in enumerate(range(0, NF, B)):
            if f0 + B < NF:
                x.read_frames_device(s, f0 + B, B, first_slot=((k + 1) % 2) * B, host_threads=T)
            s.group_get_com_batch("Solute", (k % 2) * B, B)
            s.atoms_center_batch("Solute", (k % 2) * B, B, G.Dimension.XYZ, weighted=True)
        s.sync()
        out["device_unpack_com_center_batch_%d_frames_per_s" % B] = round(NF / (time.perf_counter() - t0), 1)
    # group-limited reads (GroupXtcReader): only the 30 000 solute atoms at the head of the 5e5-atom frame are walked, copied, unpacked
    B = 64
    for w in range(2):
        x.read_frames_device(s, 0, B, first_slot=w * B, host_threads=T, group="Solute")
    s.sync()
    t0 = time.perf_counter()
    for k in range(PASSES * NF // B):
        x.read_frames_device(s, (k * B) % NF, B, first_slot=(k % 2) * B, host_threads=T, group="Solute")
    for k in range(2 * B):
        s.upload_wait(k)
    out["device_unpack_group_30k_of_%dk_batch_64_frames_per_s" % (n // 1000)] = round(PASSES * NF / (time.perf_counter() - t0), 1)
    # and out again: D2H + the library's encoder (fitted-trajectory output), T encoder threads, 32 frames per call
    x.read_frames_device(s, 0, 32, first_slot=0, host_threads=T); s.sync()
    wpath = os.path.join(tmp, "rewritten.xtc")
    with G.XtcWriter(wpath) as w:
        w.write_slots(s, 0, 32, precision=1000.0, host_threads=T)
        t0 = time.perf_counter()
        for _ in range(4):
            w.write_slots(s, 0, 32, precision=1000.0, host_threads=T)
        out["write_slots_frames_per_s"] = round(4 * 32 / (time.perf_counter() - t0), 1)
    os.remove(wpath)
    print(json.dumps(out, indent=1))
    x.close(); s.close(); os.remove(path)


if __name__ == "__main__":
    main()
