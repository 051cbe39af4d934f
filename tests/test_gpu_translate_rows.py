"""translate / wrap / centring in an orthorhombic cell as one float4 per lane (k_translate_wrap_rows, GR_TUNE_TRANSLATE_ROWS; iterators.rs:1520-1553,
atom.rs:498-545, utility.rs:109-185): the same bits as the three-rows walk it replaces -- ragged selections, atoms on faces and many cells away,
atoms without position (NaN in x: stored with NaN in y and z, left alone by both kernels) -- and the oracle as referee."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import groan_rs_amd as g
    g._lib.load()
    return g


def frames_for(n, nf, box, seed):
    rng = np.random.default_rng(seed)
    L = np.array(box[:3], np.float32)
    out = []
    for f in range(nf):
        p = (rng.uniform(-0.3, 1.3, (n, 3)) * L).astype(np.float32)
        far = rng.integers(0, n, 6); p[far] += (rng.integers(-20, 21, (6, 3)) * L).astype(np.float32)      # many cells away
        face = rng.integers(0, n, 6); p[face, rng.integers(0, 3, 6)] = 0.0                                 # on faces
        face = rng.integers(0, n, 6); ax = rng.integers(0, 3, 6); p[face, ax] = L[ax]
        nox = rng.integers(0, n, 3); p[nox, 0] = np.nan                                                    # no position
        out.append(p)
    return out


@pytest.mark.parametrize("n", [1_000, 70_001])
def test_rows_kernel_equals_the_walk_bit_for_bit(G, n):
    box = O.box_from_lengths_angles([5.0, 4.5, 6.0], [90.0, 90.0, 90.0])
    nf = 5
    frames = frames_for(n, nf, box, 3 + n)
    s = G.System(n, masses=np.ones(n, np.float32), n_slots=nf)
    s.group_create_from_ranges("mid", [(n // 7 + 1, n - n // 5 - 2)])
    s.group_create_from_ranges("head", [(0, 129)])
    res = {}
    for rows in (1, 0):
        s.set_tuning(translate_rows=rows, center_resident=0)
        got = []
        for op in ("translate_all", "wrap_mid", "translate_head", "center_mid_xz", "single_frame_wrap", "single_frame_translate_mid"):
            for f in range(nf):
                s.set_frame(frames[f], box, slot=f)
            if op == "translate_all": st = s.group_translate_batch(None, [0.7, -11.3, 3.1], 0, nf, raise_on_error=False)
            elif op == "wrap_mid": st = s.group_wrap_batch("mid", 0, nf, raise_on_error=False)
            elif op == "translate_head": st = s.group_translate_batch("head", [-0.2, 0.1, 40.0], 0, nf, raise_on_error=False)
            elif op == "center_mid_xz": st = s.atoms_center_batch("head", 0, nf, G.Dimension.XZ, raise_on_error=False)
            elif op == "single_frame_wrap":
                st = []
                for f in range(nf):
                    try: s.atoms_wrap(slot=f); st.append(0)
                    except G.GroanError as e: st.append(1)
            else:
                st = []
                for f in range(nf):
                    try: s.group_translate("mid", [1.0, 2.0, -3.0], slot=f); st.append(0)
                    except G.GroanError as e: st.append(1)
            got.append((np.array(st), [s.get_positions(f) for f in range(nf)]))
        res[rows] = got
    for (sa, pa), (sb, pb) in zip(res[1], res[0]):
        assert np.array_equal(sa, sb), (sa, sb)
        for f in range(nf):
            assert np.array_equal(pa[f], pb[f], equal_nan=True), f
    # the oracle on the whole-system translate of frame 0 (atoms with a position)
    ok = ~np.isnan(frames[0][:, 0])
    want = O.translate(frames[0], np.flatnonzero(ok), [0.7, -11.3, 3.1], box)
    assert np.array_equal(res[1][0][1][0][ok], want[ok])
    s.close()
