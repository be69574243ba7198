"""GPU: VP8 intra prediction + residual add against goldens and the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth

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


@pytest.mark.parametrize("env", [{"FFHIP_VP8_PRED_MODE": "levels"}, {"FFHIP_VP8_PRED_WAVES": "3"}, {}, {"FFHIP_VP8_PRED_SPLIT": "0"},
                                 {"FFHIP_VP8_PRED_SPLIT": "1", "FFHIP_VP8_PRED_WAVES": "1"}, {"FFHIP_VP8_PRED_SPLIT": "1", "FFHIP_VP8_PRED_WAVES": "7"}])
def test_schedulers_agree(env, monkeypatch):
    """level-synchronous launches, and the single row-form launch with few and with many waves"""
    for k, v in env.items():
        monkeypatch.setenv(k, v); capi.reload_env()
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


def test_two_streams_in_flight():
    """two batches of different geometry enqueued on two streams before either is waited for: each stream
    has its own schedule scratch, so neither call disturbs the other's counters"""
    from ffpic_amd import capi
    L = capi.require_device()
    jobs = []
    for k, (c, r, n) in enumerate(((37, 19, 3), (120, 68, 2))):
        st = L.ffhip_stream_create()
        assert st
        modes = np.stack([synth.vp8_modes(c, r, seed=800 + 10 * k + i) for i in range(n)])
        resid = np.stack([synth.vp8_residual(c * r, seed=800 + 10 * k + i) for i in range(n)])
        dm, dr = ops.DeviceBuffer(np.ascontiguousarray(modes)), ops.DeviceBuffer(np.ascontiguousarray(resid))
        ysz, csz = 256 * c * r, 64 * c * r
        planes = [ops.DeviceBuffer(nbytes=n * ysz), ops.DeviceBuffer(nbytes=n * csz), ops.DeviceBuffer(nbytes=n * csz)]
        for d in planes:
            capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, st))
        jobs.append((st, c, r, n, modes, resid, dm, dr, planes))
    for _ in range(2):      # twice each, interleaved, nothing waited for in between
        for (st, c, r, n, modes, resid, dm, dr, planes) in jobs:
            for d in planes:
                capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, st))
            capi.check(L.ffhip_vp8_predict_recon(c, r, n, modes.ctypes.data, dm.ptr, dr.ptr, c * r * 384, None,
                                                 planes[0].ptr, planes[1].ptr, planes[2].ptr, 256 * c * r, 64 * c * r, st))
    for (st, c, r, n, modes, resid, dm, dr, planes) in jobs:
        capi.check(L.ffhip_stream_sync(st))
        got = (planes[0].to_host((n, 16 * r, 16 * c), np.uint8), planes[1].to_host((n, 8 * r, 8 * c), np.uint8),
               planes[2].to_host((n, 8 * r, 8 * c), np.uint8))
        for i in range(n):
            exp = O.oracle_vp8_frame(c, r, modes[i], resid[i])
            for gp, e, name in zip(got, exp, "YUV"):
                assert np.array_equal(gp[i], e), (c, r, i, name)
        L.ffhip_stream_destroy(st)
