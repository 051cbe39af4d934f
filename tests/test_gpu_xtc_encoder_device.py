"""The device xtc encoder (gr_xtc_enc_dev.h: quantise + neighbour distances, one wave per frame walking the runs, one lane per run
emitting bits) against the host encoder (gr_xtc.h::encode_coords, itself byte-identical to the reference's writer:
tests/test_xtc_writer.py): the FILES must be equal byte for byte -- over every branch of the format: water-like runs of eight small
atoms, no runs at all, separate bit fields for huge ranges, big integers beyond 64 bits, small-range indices beyond 64 bits, atoms
without position, group writers (a block and a list), tiny and odd sizes, several frames with different headers, and a frame the format
cannot hold in the middle of a batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def _water(rng, n, span):
    """molecules of three atoms within 0.1 nm of each other, scattered over `span` nm"""
    o = rng.uniform(0, span, ((n + 2) // 3, 3))
    p = np.repeat(o, 3, axis=0)[:n] + rng.normal(0, 0.05, (n, 3))
    return p.astype(np.float32)


def _walk(rng, n, lo, hi):
    """a random walk whose steps have components in +-hi and an L1 length of at least lo (in nm): every atom is 'near' its predecessor
    only for a large small-range index"""
    steps = rng.uniform(-hi, hi, (n, 3))
    short = np.abs(steps).sum(1) < lo
    steps[short] *= (lo / np.abs(steps[short]).sum(1))[:, None] * 1.01
    return np.cumsum(steps, axis=0).astype(np.float32)


CASES = {
    # name: (n atoms, frames, precision, generator)
    "water": (30_011, 5, 1000.0, lambda r, n, f: _water(r, n, 9.0 + f)),
    "gas (no atom near its neighbour)": (20_000, 3, 1000.0, lambda r, n, f: r.uniform(0, 30.0, (n, 3)).astype(np.float32)),
    "chain of small steps": (25_003, 3, 1000.0, lambda r, n, f: np.cumsum(r.normal(0, 0.02 * (f + 1), (n, 3)), axis=0).astype(np.float32)),
    "separate bit fields (range > 2^24 quanta)": (5_000, 2, 1.0e6, lambda r, n, f: _water(r, n, 25.0)),
    "big integer beyond 64 bits": (5_000, 2, 1000.0, lambda r, n, f: _water(r, n, 6000.0)),
    "small-range index beyond 64 bits": (4_000, 2, 1000.0, lambda r, n, f: _walk(r, n, 2700.0, 1500.0)),
    "ten atoms": (10, 4, 1000.0, lambda r, n, f: _water(r, n, 3.0)),
    "257 atoms (one past a window)": (257, 3, 1000.0, lambda r, n, f: _water(r, n, 4.0)),
    "identical atoms": (1_000, 2, 1000.0, lambda r, n, f: np.full((n, 3), 1.234, np.float32)),
    "a protein in water (one long chain of near atoms: no place where a run must start)": (60_000, 2, 1000.0, lambda r, n, f: np.concatenate(
        [_water(r, 20_001, 8.0), (4.0 + np.cumsum(r.normal(0, 0.07, (19_998, 3)), axis=0)).astype(np.float32), _water(r, 20_001, 8.0)])),
    "mixed: water, gas, water": (40_000, 2, 500.0, lambda r, n, f: np.concatenate([_water(r, 15_000, 8.0), r.uniform(0, 8.0, (10_000, 3)).astype(np.float32), _water(r, 15_000, 8.0)])),
}


def _write(G, tmp_path, tag, frames, box, device, precision, group=None, blocks=None, min_total=0):
    n, nf = frames[0].shape[0], len(frames)
    s = G.System(n, n_slots=nf)
    s.set_tuning(xtc_device_encode=device)
    for f in range(nf):
        s.set_frame(frames[f], box, slot=f)
    if group:
        s.group_create_from_ranges(group, blocks)
    path = tmp_path / ("%s_%d.xtc" % (tag, device))
    err = None
    with G.XtcWriter(path) as w:
        try:
            w.write_slots(s, 0, nf, group=group, steps=np.arange(nf, dtype=np.int64) * 10, times=np.arange(nf, dtype=np.float32) * 0.5, precision=precision, host_threads=2)
        except G.XtcError as e:
            err = e.status
    took = s.stat("xtc_device_frames")
    s.close()
    return open(path, "rb").read(), err, took


@pytest.mark.parametrize("case", list(CASES))
def test_device_encoder_writes_the_host_encoders_bytes(G, tmp_path, case):
    n, nf, precision, gen = CASES[case]
    rng = np.random.default_rng(list(CASES).index(case))
    frames = [gen(rng, n, f) for f in range(nf)]
    if n > 1000:                                            # atoms without position: written as the origin
        for f in range(nf):
            frames[f][rng.integers(0, n, 7)] = np.nan
    box = np.array([9, 9, 9, 0, 0, 0, 0, 0, 0], np.float32)
    if n * nf < 200_000:                                    # (the library's own threshold would keep outputs this small on the host)
        frames = frames * (200_000 // (n * nf) + 1)
    dev, e1, took = _write(G, tmp_path, "c", frames, box, 1, precision)
    host, e0, took0 = _write(G, tmp_path, "c", frames, box, 0, precision)
    assert e1 is None and e0 is None and took == len(frames) and took0 == 0
    assert len(dev) == len(host) and dev == host, (case, len(dev), len(host), next(i for i in range(min(len(dev), len(host))) if dev[i] != host[i]))
    # the case does reach the branch its name says (header of the first frame: minint, maxint, smallidx behind the 56 + 4 bytes)
    hd = np.frombuffer(dev[60:88], dtype=">i4").astype(np.int64)
    size = hd[3:6] - hd[0:3] + 1
    if "separate bit fields" in case: assert (size > 0xffffff).any()
    if "beyond 64 bits" in case and "big" in case: assert (size <= 0xffffff).all() and int(size[0]) * int(size[1]) * int(size[2]) >= 2 ** 64
    if "small-range index" in case: assert hd[6] > 64, hd[6]


@pytest.mark.parametrize("blocks", [[(5_000, 24_999)], [(i, i + 2) for i in range(3, 59_990, 6)]])
def test_group_writers_on_the_device(G, tmp_path, blocks):
    rng = np.random.default_rng(3)
    n, nf = 60_000, 12
    frames = [_water(rng, n, 10.0) for _ in range(nf)]
    box = np.array([10, 10, 10, 0, 0, 0, 0, 0, 0], np.float32)
    dev, e1, took = _write(G, tmp_path, "g", frames, box, 1, 1000.0, group="S", blocks=blocks)
    host, e0, _ = _write(G, tmp_path, "g", frames, box, 0, 1000.0, group="S", blocks=blocks)
    assert e1 is None and e0 is None and took == nf and dev == host


def test_a_frame_the_format_cannot_hold_in_the_middle_of_a_device_batch(G, tmp_path):
    rng = np.random.default_rng(8)
    n, nf = 50_000, 6
    frames = [_water(rng, n, 7.0) for _ in range(nf)]
    frames[3][777, 1] = 3.0e6                               # its quantum does not fit 32 bits
    box = np.array([7, 7, 7, 0, 0, 0, 0, 0, 0], np.float32)
    dev, e1, took = _write(G, tmp_path, "o", frames, box, 1, 1000.0)
    host, e0, _ = _write(G, tmp_path, "o", frames, box, 0, 1000.0)
    assert e1 == 9 and e0 == 9 and took == 3 and dev == host and len(dev) > 0      # GR_E_OUT_OF_RANGE; the three frames before it are in the file
    frames[3][777, 1] = 1.0; frames[5][:, 0] += np.linspace(0, 4.0e6, n, dtype=np.float32)      # range wider than the integers: refused as well
    dev, e1, took = _write(G, tmp_path, "o2", frames, box, 1, 1000.0)
    host, e0, _ = _write(G, tmp_path, "o2", frames, box, 0, 1000.0)
    assert e1 == 9 and e0 == 9 and took == 5 and dev == host


def test_randomised_systems(G, tmp_path):
    """mixtures of runs, jumps and scales, random sizes: the two encoders agree on every byte"""
    rng = np.random.default_rng(11)
    for k in range(40):
        n = int(rng.integers(10, 20_000))
        nf = max(1, 200_000 // n + 1)
        scale = float(10.0 ** rng.uniform(-2, 3))
        precision = float(rng.choice([10.0, 100.0, 1000.0, 12345.0]))
        frames = []
        for f in range(min(nf, 6)):
            parts, left = [], n
            while left > 0:
                m = int(min(left, rng.integers(1, 400)))
                kind = int(rng.integers(0, 3))
                if kind == 0: parts.append(_water(rng, m, scale))
                elif kind == 1: parts.append(rng.uniform(-scale, scale, (m, 3)).astype(np.float32))
                else: parts.append((rng.uniform(0, scale, (1, 3)) + np.cumsum(rng.normal(0, scale * 10.0 ** rng.uniform(-4, -1), (m, 3)), axis=0)).astype(np.float32))
                left -= m
            frames.append(np.concatenate(parts)[:n])
        frames = (frames * (nf // len(frames) + 1))[:nf]
        box = np.array([9, 9, 9, 0, 0, 0, 0, 0, 0], np.float32)
        dev, e1, took = _write(G, tmp_path, "r%d" % k, frames, box, 1, precision)
        host, e0, _ = _write(G, tmp_path, "r%d" % k, frames, box, 0, precision)
        assert e1 == e0 and dev == host, (k, n, nf, scale, precision, e1, e0)


def test_a_small_group_written_from_more_slots_than_a_grid_has_rows(G, tmp_path):
    """12 atoms of a ligand out of 66 000 resident frames: the device encoder takes its frames as the y dimension of its grids, which
    ends at 65 535 -- the rounds must be cut there (the launch failed with hipErrorInvalidConfiguration, the call returned
    GR_E_HIP and the host encoders were never tried).  Same bytes as the host encoders."""
    n, nf = 300, 66_000
    box = np.array([6, 6, 6, 0, 0, 0, 0, 0, 0], np.float32)
    out = {}
    for device in (1, 0):
        s = G.System(n, n_slots=nf + 1)
        s.set_tuning(xtc_device_encode=device)
        s.synth_reference(nf, box, 1.0, 11)
        s.synth_frames(nf, 0, nf, 0, 0.05, 11)
        s.group_create_from_ranges("L", [(10, 21)])
        path = tmp_path / ("many_%d.xtc" % device)
        with G.XtcWriter(path) as w:
            w.write_slots(s, 0, nf, group="L", steps=np.arange(nf, dtype=np.int64), times=np.arange(nf, dtype=np.float32), precision=1000.0, host_threads=4)
        out[device] = (open(path, "rb").read(), s.stat("xtc_device_frames"))
        s.close()
    assert out[1][1] == nf and out[0][1] == 0
    assert out[1][0] == out[0][0] and len(out[0][0]) > nf * 60


def test_a_frame_that_is_one_dense_chain_goes_to_the_host_encoders(G, tmp_path):
    """a protein-only output has no atom far from its predecessor: no place where a run MUST start, so the device planner would walk the
    whole frame with one thread.  The first round declines such frames and the host encoders write the call: same bytes,
    `xtc_device_frames` stays 0; a chain that is a third of the frame (the case above) still runs on the device."""
    rng = np.random.default_rng(21)
    n, nf = 40_000, 6
    def chain():    # bonded neighbours: every step ~0.1 nm long (so the frame's smallest step, which sets the range of the small-atom index, is large and
        u = rng.normal(0, 1, (n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]      # `larger` -- the reach of a run -- covers every step)
        return (25.0 + np.cumsum(u * rng.uniform(0.09, 0.11, (n, 1)), axis=0)).astype(np.float32)
    frames = [chain() for _ in range(nf)]
    box = np.array([50, 50, 50, 0, 0, 0, 0, 0, 0], np.float32)
    dev, e1, took = _write(G, tmp_path, "chain", frames, box, 1, 1000.0)
    host, e0, _ = _write(G, tmp_path, "chain", frames, box, 0, 1000.0)
    assert e1 is None and e0 is None and took == 0 and dev == host and len(dev) > 0
