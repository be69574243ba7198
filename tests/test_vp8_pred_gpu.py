"""GPU: VP8 intra prediction + residual add against goldens and the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import ops, synth

pytestmark = pytest.mark.gpu


def check(c, r, modes, resid, resmap=None):
    got = ops.vp8_predict_recon(c, r, modes[None], resid[None], None if resmap is None else resmap[None])
    exp = O.oracle_vp8_frame(c, r, modes, resid, resmap)
    for g, e, name in zip(got, exp, "YUV"):
        assert np.array_equal(g[0], e), name


def test_golden_frames(golden):
    g = golden("vp8_frames.npz")
    for tag in "abc":
        c, r = [int(x) for x in g[f"{tag}_dims"]]
        got = ops.vp8_predict_recon(c, r, g[f"{tag}_modes"][None], g[f"{tag}_residual"][None], g[f"{tag}_resmap"][None])
        assert np.array_equal(got[0][0], g[f"{tag}_y"]) and np.array_equal(got[1][0], g[f"{tag}_u"]) and \
            np.array_equal(got[2][0], g[f"{tag}_v"]), tag
    for ym in range(4):      # includes V_PRED on the top row and H_PRED in the left column (raw-memory reads)
        got = ops.vp8_predict_recon(4, 3, g[f"m{ym}_modes"][None], g[f"m{ym}_residual"][None])
        assert np.array_equal(got[0][0], g[f"m{ym}_y"]) and np.array_equal(got[1][0], g[f"m{ym}_u"]), ym
    for bm in range(10):
        got = ops.vp8_predict_recon(4, 3, g[f"b{bm}_modes"][None], g[f"b{bm}_residual"][None])
        assert np.array_equal(got[0][0], g[f"b{bm}_y"]), bm


@pytest.mark.parametrize("c,r", [(1, 1), (2, 1), (1, 4), (3, 2), (17, 9), (40, 30)])
def test_frames_vs_oracle(c, r):
    modes = synth.vp8_modes(c, r, seed=c * 100 + r)
    resid = synth.vp8_residual(c * r, seed=c * 100 + r, amplitude=90)
    check(c, r, modes, resid)


def test_batch_and_resmap():
    c, r, n = 9, 6, 5
    modes = np.stack([synth.vp8_modes(c, r, seed=70 + i) for i in range(n)])
    resid = np.stack([synth.vp8_residual(c * r, seed=70 + i) for i in range(n)])
    rm = np.tile(np.arange(c * r, dtype=np.int32), (n, 1))
    rm[:, 7] = 6
    rm[2, 30:34] = 29
    got = ops.vp8_predict_recon(c, r, modes, resid, rm)
    for i in range(n):
        exp = O.oracle_vp8_frame(c, r, modes[i], resid[i], rm[i])
        for gp, e in zip(got, exp):
            assert np.array_equal(gp[i], e), i


def test_1080p_frame():
    """BASELINE config 4 geometry: 120 x 68 macroblocks"""
    c, r = 120, 68
    check(c, r, synth.vp8_modes(c, r, seed=4), synth.vp8_residual(c * r, seed=4))
