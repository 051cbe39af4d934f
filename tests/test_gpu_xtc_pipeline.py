"""config[4]-style pipeline on the GPU box: host xtc decode (the library's decoder, one frame per thread, straight into
pinned buffers) overlapped with H2D on the copy stream and the GPU COM + centre/wrap work, double-buffered.
The 5e5-atom truncated-octahedron trajectory is written at run time with the library's own encoder; its sha256 is pinned to
the file the reference's xdrfile writer produces from the same coordinates (tests/golden/xtc_pins.json).  Parity: per-frame COM and wrapped coordinates against the oracle; throughput and the stage times
go to gpurun_out/xtc_pipeline.json."""
import json
import os
import queue
import threading
import time

import numpy as np
import pytest

import oracle_lib as O
import xtc_cases as XC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decode_upload_com_wrap_pipeline(tmp_path):
    import groan_rs_amd as G
    n, n_frames, n_threads, n_buf = 500_000, 32, 8, 8
    frames, box9, boxm = XC.octahedron_case(n)                                      # truncated octahedron (simbox.rs:329-342)
    prot = slice(0, 30_000)                                                        # a compact "solute" to take the COM of
    path = tmp_path / "octa_5e5.xtc"
    t0 = time.time()
    XC.write_own(G, path, [frames[f % 4] for f in range(n_frames)], boxm, 1000.0)
    t_write = time.time() - t0
    assert XC.sha256_file(path) == XC.pin("octahedron_5e5_x32")["sha256"]           # = the reference writer's bytes
    masses = np.array([15.999, 1.008, 1.008], np.float32)[np.arange(n) % 3]

    x = G.XtcFile(path)
    assert x.n_atoms == n and x.n_frames == n_frames
    sysd = G.System(n, masses=masses, n_slots=n_buf)
    sysd.group_create_from_ranges("Solute", [(0, 29_999)])
    staging = [G.pinned_array((n, 3)) for _ in range(n_buf)]

    # ---- stage times alone
    t0 = time.perf_counter()
    for f in range(8):
        x.read_frame(f, out=staging[f % n_buf][0])
    t_dec1 = (time.perf_counter() - t0) / 8                                          # one thread, seconds per frame

    # ---- pipeline: decoder threads fill pinned buffers; the main thread uploads (copy stream) and runs the GPU work
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
                if f >= n_frames:
                    return
                next_frame[0] += 1
            b = free_q.get()
            _, box, step, tm, _ = x.read_frame(f, out=staging[b][0])
            with cond:
                ready[f] = (b, box)
                cond.notify_all()

    ths = [threading.Thread(target=decoder) for _ in range(n_threads)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    coms, wrapped_first = [], None
    pending = None
    for f in range(n_frames):
        with cond:
            while f not in ready:
                cond.wait()
            b, box = ready.pop(f)
        slot = f % n_buf
        sysd.upload_async(staging[b][0], box, slot)                                 # H2D on the copy stream
        if pending is not None:                                                     # the previous staging buffer is free once its copy is done
            pb, pslot = pending
            sysd.upload_wait(pslot); free_q.put(pb)
        pending = (b, slot)
        coms.append(sysd.group_get_com("Solute", slot=slot))                        # GPU: Bai-Breen + unwrap + COM
        sysd.atoms_center_mass("Solute", G.Dimension.XYZ, slot=slot)                # GPU: centre on the solute + wrap everything
        if f == 1:
            wrapped_first = sysd.get_positions(slot)
    sysd.sync()
    t_all = time.perf_counter() - t0
    [t.join() for t in ths]

    # ---- parity against the oracle on the decoded frames (decode is bit-exact, see tests/test_xtc_decoder.py)
    idx = np.arange(30_000)
    O.set_accumulate_f64(True)
    try:
        for f in (0, 1, 5, 31):
            dec = x.read_frame(f)[0]
            want = O.get_center(dec, idx, box9, mass=masses)
            assert np.abs(coms[f] - want).max() <= 1e-5, (f, coms[f], want)
        dec = x.read_frame(1)[0]
        want = O.atoms_center(dec, idx, "xyz", box9, mass=masses)
        assert np.abs(wrapped_first - want).max() <= 2e-5
    finally:
        O.set_accumulate_f64(False)
    out = {"n_atoms": n, "n_frames": n_frames, "decode_threads": n_threads, "file_MB": round(os.path.getsize(path) / 1e6, 1),
           "decode_one_thread_frames_per_s": round(1.0 / t_dec1, 1), "pipeline_frames_per_s": round(n_frames / t_all, 1),
           "pipeline_wall_s": round(t_all, 3), "own_writer_one_thread_s": round(t_write, 2),
           "stages": "xtc decode (host threads) || H2D copy stream || group_get_com + atoms_center_mass (all atoms) on the GPU"}
    # ---- the same pipeline with the frames unpacked ON THE DEVICE: host skims the framing (n_threads workers), the
    # compressed stream crosses PCIe, k_xtc_unpack decodes a batch of frames in one launch on the copy stream
    B = n_buf
    t0 = time.perf_counter()
    for f0 in range(0, 2 * B, B):                                                    # warm-up (staging buffers, first launch)
        x.read_frames_device(sysd, f0, B, first_slot=0, host_threads=n_threads)
    sysd.sync()
    t_skim = time.perf_counter()
    for f0 in range(0, n_frames, B):
        x.read_frames_device(sysd, f0, B, first_slot=0, host_threads=n_threads)
    for s_ in range(B):
        sysd.upload_wait(s_)
    t_unpack_only = (time.perf_counter() - t_skim) / n_frames                        # skim + H2D + unpack, no analysis
    coms_dev, wrapped_dev = [], None
    t0 = time.perf_counter()
    for f0 in range(0, n_frames, B):
        x.read_frames_device(sysd, f0, B, first_slot=0, host_threads=n_threads)
        for k in range(B):
            coms_dev.append(sysd.group_get_com("Solute", slot=k))
            sysd.atoms_center_mass("Solute", G.Dimension.XYZ, slot=k)
            if f0 + k == 1:
                wrapped_dev = sysd.get_positions(k)
    sysd.sync()
    t_dev = time.perf_counter() - t0
    assert np.array_equal(np.array(coms_dev), np.array(coms))                        # same decoded bits -> same results
    assert np.array_equal(wrapped_dev, wrapped_first)
    # ---- the same with the per-frame analyses batched too (gr_group_center_batch / gr_atoms_center_batch)
    coms_b = []
    t0 = time.perf_counter()
    for f0 in range(0, n_frames, B):
        x.read_frames_device(sysd, f0, B, first_slot=0, host_threads=n_threads)
        cb, _ = sysd.group_get_com_batch("Solute", 0, B)
        coms_b.append(cb)
        sysd.atoms_center_batch("Solute", 0, B, G.Dimension.XYZ, weighted=True)
    sysd.sync()
    t_devb = time.perf_counter() - t0
    assert np.array_equal(np.concatenate(coms_b), np.array(coms))
    out["device_unpack_batched_analysis_frames_per_s"] = round(n_frames / t_devb, 1)
    # ---- and out again: D2H + the library's encoder (fitted-trajectory output, NEXT-4), n_threads encoders
    wpath = tmp_path / "rewritten.xtc"
    x.read_frames_device(sysd, 0, B, first_slot=0, host_threads=n_threads)
    with G.XtcWriter(wpath) as w:
        w.write_slots(sysd, 0, B, precision=1000.0, host_threads=n_threads)          # warm-up
        t0 = time.perf_counter()
        for _ in range(3):
            w.write_slots(sysd, 0, B, precision=1000.0, host_threads=n_threads)
        t_write_ours = (time.perf_counter() - t0) / (3 * B)
    y = G.XtcFile(wpath)
    assert y.n_frames == 4 * B and np.array_equal(y.read_frame(B + 1)[0], x.read_frame(1)[0])   # decode(encode(decode)) is a fixed point
    y.close()
    out["write_slots_frames_per_s"] = round(1.0 / t_write_ours, 1)
    out["device_unpack_pipeline_frames_per_s"] = round(n_frames / t_dev, 1)
    out["device_unpack_only_frames_per_s"] = round(1.0 / t_unpack_only, 1)
    out["device_unpack_batch"] = B
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "xtc_pipeline.json"), "w"), indent=1)
    print(out)
    for _, ptr in staging:
        G.pinned_free(ptr)
    x.close(); sysd.close()
