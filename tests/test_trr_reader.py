"""The library's own trr reader (groan_rs_amd/csrc/gr_trr.h behind gr_trr_*) against the reference's reader: every frame of
the reference's small trr data files must decode to the bits the vendored C xdrfile's read_trr produces (expected digests:
tests/golden/trr_expected.json from tests/golden/make_trr_expected.py over oracle/_ref) -- positions, velocities, forces,
step / time / lambda / box, single and double precision, frames that lack sections.  Host only: no GPU needed.
Reference: TrrFrameData::from_frame / update_system (src/io/trr_io.rs:49-135), its tests :574-1330."""
import hashlib
import json
import os

import numpy as np
import pytest

import groan_rs_amd as G

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXP = json.load(open(os.path.join(GOLD, "trr_expected.json")))


def rows_to_box9(m):
    return [m[0], m[4], m[8], m[1], m[2], m[3], m[5], m[6], m[7]]


@pytest.mark.parametrize("name", sorted(EXP))
def test_every_frame_like_the_reference_reader(name):
    t = G.TrrFile(os.path.join(GOLD, name))
    e = EXP[name]
    assert t.n_atoms == e["n_atoms"] and t.n_frames == len(e["frames"])
    for k, fr in enumerate(e["frames"]):
        x, v, f, box9, step, time, lam = t.read_frame(k, velocities=True, forces=True)
        assert hashlib.sha256(x.tobytes()).hexdigest() == fr["x"], (name, k)
        assert hashlib.sha256(v.tobytes()).hexdigest() == fr["v"] and hashlib.sha256(f.tobytes()).hexdigest() == fr["f"]
        assert step == fr["step"] and np.float32(time) == np.float32(fr["time"]) and np.float32(lam) == np.float32(fr["lambda"])
        info = t.frame_info(k)
        # (non-zero data needs its section; a section may also be present and all zero)
        assert (info["positions"] or not fr["has_x"]) and (info["velocities"] or not fr["has_v"]) and (info["forces"] or not fr["has_f"])
        assert info["double_precision"] == ("double" in name)
        assert np.array_equal(np.float32(box9), np.float32(rows_to_box9(fr["box_rowmajor"])))
    t.close()


REF_TF = "/root/reference/test_files"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_TF, "short_trajectory.trr")), reason="reference tree not mounted (build container only)")
def test_known_answers_of_the_reference_tests():
    """the values the reference pins for its 16 844-atom trajectories (src/io/trr_io.rs:574-640, 806-850); the files are too
    large for fixtures, so this runs where the reference tree is mounted"""
    t = G.TrrFile(os.path.join(REF_TF, "short_trajectory.trr"))
    assert t.n_atoms == 16844
    x, v, f, box9, step, time, lam = t.read_frame(0, velocities=True, forces=True)
    assert step == 0 and lam == 0.0 and abs(time) < 1e-6
    np.testing.assert_allclose(box9[:3], [13.01331, 13.01331, 11.25347], atol=2e-6)
    np.testing.assert_allclose(x[0], [9.497, 1.989, 7.498], atol=2e-6); np.testing.assert_allclose(v[0], [-0.0683, 0.1133, 0.0005], atol=2e-6)
    np.testing.assert_allclose(f[0], [-6.2916107, -276.57983, -306.23727], rtol=1e-6)
    np.testing.assert_allclose(x[16843], [8.829, 11.186, 2.075], atol=2e-6); np.testing.assert_allclose(v[16843], [0.0712, 0.2294, -0.1673], atol=2e-6)
    np.testing.assert_allclose(f[16843], [-21.009035, -6.7285156, -68.827545], rtol=1e-6)
    _, _, _, box9, step, time, _ = t.read_frame(1)
    assert step == 6000 and abs(time - 120.0) < 1e-4
    np.testing.assert_allclose(box9[:3], [13.024242, 13.024242, 11.242146], atol=2e-6)
    t.close()
    if not os.path.exists(os.path.join(REF_TF, "short_trajectory_double.trr")):    # (not in every checkout of the reference)
        return
    d = G.TrrFile(os.path.join(REF_TF, "short_trajectory_double.trr"))
    x, v, f, box9, step, time, lam = d.read_frame(0, velocities=True, forces=True)
    assert d.frame_info(0)["double_precision"] and step == 0
    np.testing.assert_allclose(x[0], [9.497161, 1.9891102, 7.497941], rtol=1e-6); np.testing.assert_allclose(v[0], [-0.06389237, 0.054320477, 0.008154817], rtol=1e-6)
    np.testing.assert_allclose(f[0], [-6.330056, -278.8763, -305.94952], rtol=1e-6)
    d.close()


def test_frames_iterator_marks_missing_positions():
    """TrrFrameData::update_system (trr_io.rs:108-112): an all-zero position is "no position" -- whether the frame has no
    position section or carries zeros in it"""
    t = G.TrrFile(os.path.join(GOLD, "dodecahedron_trajectory_full.trr"))
    seen = {"some": 0, "none": 0}
    for k, (x, box, step, time) in enumerate(t.frames()):
        raw = t.read_frame(k)[0]
        zero = ~raw.any(axis=1)
        assert np.array_equal(np.isnan(x[:, 0]), zero) and np.array_equal(x[~zero], raw[~zero])
        if not t.frame_info(k)["positions"]:
            assert zero.all()
        seen["none" if zero.all() else "some"] += 1
    assert k + 1 == t.n_frames and seen["some"] > 0 and seen["none"] > 0
    t.close()


@pytest.mark.parametrize("mutation", ["truncate_header", "truncate_payload", "bad_magic", "bad_string", "bad_sizes", "empty", "text"])
def test_corrupt_files_are_rejected_not_crashed(tmp_path, mutation):
    raw = bytearray(open(os.path.join(GOLD, "triclinic_trajectory.trr"), "rb").read())
    if mutation == "truncate_header": raw = raw[:40]
    elif mutation == "truncate_payload": raw = raw[:len(raw) - 100]
    elif mutation == "bad_magic": raw[3] ^= 0xff
    elif mutation == "bad_string": raw[14] ^= 0x20
    elif mutation == "bad_sizes": raw[24 + 4 * 7 + 3] ^= 0x04            # x_size no longer 12 * natoms
    elif mutation == "empty": raw = bytearray()
    elif mutation == "text": raw = bytearray(b"Not a trr file.")          # the reference's fake_trr.trr
    p = tmp_path / "bad.trr"
    p.write_bytes(bytes(raw))
    if mutation == "empty":
        t = G.TrrFile(p); assert t.n_frames == 0; t.close()
    else:
        with pytest.raises(G.XtcError):
            G.TrrFile(p)


REF_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libxdrfile_ref.so")


@pytest.mark.parametrize("seed", range(6))
def test_writer_produces_the_reference_writers_bytes(tmp_path, seed):
    """TrrWriter (src/io/trr_io.rs:441-520): random frames written by the library and by the reference's vendored C write_trr
    (oracle/_ref) are the same bytes; the library's reader gets them back unchanged"""
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    import ctypes as C
    lib = C.CDLL(REF_SO)
    lib.xdrfile_open.restype = C.c_void_p; lib.xdrfile_open.argtypes = [C.c_char_p, C.c_char_p]
    lib.xdrfile_close.argtypes = [C.c_void_p]
    lib.write_trr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(88 + seed)
    n = int(rng.choice([1, 3, 10, 50, 777]))
    frames = [(rng.normal(0, 5, (n, 3)).astype(np.float32), rng.normal(0, 1, (n, 3)).astype(np.float32), rng.normal(0, 300, (n, 3)).astype(np.float32)) for _ in range(4)]
    boxm = np.array([[7.5, 0, 0], [0, 6.5, 0], [1.0, 2.0, 5.5]], np.float32)
    box9 = np.array(rows_to_box9(boxm.reshape(-1).tolist()), np.float32)
    refp, ours = tmp_path / "ref.trr", tmp_path / "ours.trr"
    fh = lib.xdrfile_open(str(refp).encode(), b"w")
    for i, (x, v, f) in enumerate(frames):
        assert lib.write_trr(fh, n, i * 100, C.c_float(0.5 * i), C.c_float(0.25), boxm.ctypes.data, x.ctypes.data, v.ctypes.data, f.ctypes.data) == 0
    lib.xdrfile_close(fh)
    with G.TrrWriter(ours) as w:
        for i, (x, v, f) in enumerate(frames):
            w.write_frame(x, box9, step=i * 100, time=0.5 * i, lambda_=0.25, velocities=v, forces=f)
    assert open(ours, "rb").read() == open(refp, "rb").read()
    t = G.TrrFile(ours)
    assert t.n_frames == 4 and t.n_atoms == n
    for i, (x, v, f) in enumerate(frames):
        gx, gv, gf, gb, step, time, lam = t.read_frame(i, velocities=True, forces=True)
        assert np.array_equal(gx, x) and np.array_equal(gv, v) and np.array_equal(gf, f) and np.array_equal(gb, box9)
        assert step == i * 100 and time == np.float32(0.5 * i) and lam == 0.25
    t.close()
    # the default: no velocities / forces -> zero sections (what the reference writes for atoms that have none); NaN x -> zeros
    x = frames[0][0].copy(); x[0, 0] = np.nan
    with G.TrrWriter(tmp_path / "z.trr") as w:
        w.write_frame(x, box9)
    t = G.TrrFile(tmp_path / "z.trr")
    gx, gv, gf, _, _, _, _ = t.read_frame(0, velocities=True, forces=True)
    assert not gx[0].any() and np.array_equal(gx[1:], x[1:]) and not gv.any() and not gf.any() and t.frame_info(0)["velocities"]
    t.close()
