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


@pytest.mark.parametrize("env", [{"FFHIP_VP8_PRED_MODE": "levels"}, {"FFHIP_VP8_PRED_WAVES": "3"}, {}])
def test_schedulers_agree(env, monkeypatch):
    """level-synchronous launches, and the single row-form launch with few and with many waves"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c, r, n = 23, 11, 4
    modes = np.stack([synth.vp8_modes(c, r, seed=170 + i) for i in range(n)])
    modes[1, np.arange(r) * c, 0] = 3        # H_PRED down the whole left column: every row waits for the full row above
    resid = np.stack([synth.vp8_residual(c * r, seed=170 + i, amplitude=60) for i in range(n)])
    got = ops.vp8_predict_recon(c, r, modes, resid)
    for i in range(n):
        exp = O.oracle_vp8_frame(c, r, modes[i], resid[i])
        for gp, e, name in zip(got, exp, "YUV"):
            assert np.array_equal(gp[i], e), (env, i, name)


def test_row_handoff_stress():
    """many 1080p frames at once, twice over the same planes: every byte of every frame against the
    oracle -- the row-to-row hand-off (agent-scope stores, progress counters) under uneven load"""
    c, r, n = 120, 68, 6
    modes = np.stack([synth.vp8_modes(c, r, seed=300 + i) for i in range(n)])
    resid = np.stack([synth.vp8_residual(c * r, seed=300 + i, amplitude=70) for i in range(n)])
    exp = [O.oracle_vp8_frame(c, r, modes[i], resid[i]) for i in range(n)]
    for _ in range(2):
        got = ops.vp8_predict_recon(c, r, modes, resid)
        for i in range(n):
            for gp, e, name in zip(got, exp[i], "YUV"):
                assert np.array_equal(gp[i], e), (i, name)
