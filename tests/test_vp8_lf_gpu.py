"""GPU: VP8 in-loop filter against goldens and the oracle, and the whole
WebP post-entropy chain residual -> predict -> loop filter -> colour against the oracle chain."""
import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth

pytestmark = pytest.mark.gpu


def oracle_lf(c, r, ft, modes, filt, planes):
    p = [np.ascontiguousarray(x).copy() for x in planes]
    O.ffo().ffo_vp8_loopfilter_frame(c, r, ft, np.ascontiguousarray(modes).reshape(-1), np.ascontiguousarray(filt).reshape(-1),
                                     p[0].reshape(-1), p[1].reshape(-1), p[2].reshape(-1))
    return p


def test_golden(golden):
    g = golden("vp8_loopfilter.npz")
    for tag in "ab":
        c, r = [int(x) for x in g[f"{tag}_dims"]]
        for ft in (1, 2):
            got = ops.vp8_loopfilter(c, r, ft, g[f"{tag}_modes"][None], g[f"{tag}_filters"], g[f"{tag}_y"][None],
                                     g[f"{tag}_u"][None], g[f"{tag}_v"][None])
            for k, pl in zip("yuv", got):
                assert np.array_equal(pl[0], g[f"{tag}_f{ft}_{k}"]), (tag, ft, k)


@pytest.mark.parametrize("c,r,ft", [(1, 1, 1), (1, 1, 2), (2, 3, 2), (17, 9, 1), (17, 9, 2), (120, 68, 2)])
def test_vs_oracle(c, r, ft):
    rng = np.random.default_rng(c * 10 + r)
    modes = synth.vp8_modes(c, r, seed=c)
    modes[:, 18] = rng.integers(0, 4, size=c * r)
    filt = synth.vp8_filters(seed=c + r)
    base = synth.vp8_blocky_planes(c, r, seed=r)
    got = ops.vp8_loopfilter(c, r, ft, modes[None], filt, base[0][None], base[1][None], base[2][None])
    exp = oracle_lf(c, r, ft, modes, filt, base)
    for gp, e in zip(got, exp):
        assert np.array_equal(gp[0], e)
    if ft == 0:
        assert np.array_equal(got[0][0], base[0])


def test_webp_chain_batch():
    """levels -> residual -> intra predict/recon -> loop filter -> BGRA for 3 frames, every stage on
    the GPU, against the same chain of oracle functions"""
    c, r, n = 10, 7, 3
    n_mb = c * r
    q = synth.vp8_quant(seed=2)
    filt = synth.vp8_filters(seed=2)
    frames = []
    for i in range(n):
        lv, info = synth.vp8_macroblocks(n_mb, seed=100 + i)
        modes = synth.vp8_modes(c, r, seed=100 + i)
        modes[:, 18] = info[:, 26]
        info[:, 25] = modes[:, 0] != 4            # Y2 exactly when the MB is not B_PRED
        frames.append((lv, info, modes))
    res_gpu = [ops.vp8_residual_batch(lv, info, q) for lv, info, _ in frames]
    modes = np.stack([f[2] for f in frames])
    y, u, v = ops.vp8_predict_recon(c, r, modes, np.stack(res_gpu))
    y, u, v = ops.vp8_loopfilter(c, r, 2, modes, filt, y, u, v)
    bgra = ops.yuv420_to_bgra(y, u, v, r, c)
    F = O.ffo()
    for i, (lv, info, m) in enumerate(frames):
        res = np.zeros((n_mb, 384), np.int16)
        for k in range(n_mb):
            F.ffo_vp8_residual_mb(np.ascontiguousarray(lv[k]).reshape(-1), info[k], int(info[k, 25]),
                                  np.ascontiguousarray(q[info[k, 26], :6]), res[k])
        planes = O.oracle_vp8_frame(c, r, m, res)
        planes = oracle_lf(c, r, 2, m, filt, planes)
        out = np.zeros((16 * r, 16 * c * 4), np.uint8)
        F.ffo_yuv420_to_bgra32(out.reshape(-1), 16 * c * 4, planes[0].reshape(-1), planes[1].reshape(-1),
                               planes[2].reshape(-1), 16 * c, 8 * c, r, c)
        assert np.array_equal(bgra[i], out), i


def test_webp_file_config4(golden):
    """the reference's whole-file decode of a real lossy WebP, reproduced on the GPU from the
    per-macroblock dump of its own decoder (tests/golden/make_golden.py::gen_webp_file)"""
    g = golden("webp_file.npz")
    w, h, pitch = [int(x) for x in g["dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    y, u, v = ops.vp8_predict_recon(c, r, g["modes"][None], g["residual"][None])
    y, u, v = ops.vp8_loopfilter(c, r, 0, g["modes"][None], synth.vp8_filters(), y, u, v)   # level 0: a no-op
    bgra = ops.yuv420_to_bgra(y, u, v, r, c, pitch=pitch)
    assert np.array_equal(bgra[0][:h], g["bgra"])


@pytest.mark.parametrize("tag", ["q55", "q40"])
def test_webp_file_with_loop_filter(golden, tag):
    """f3 at file level on the GPU: predict + reconstruct -> ffhip_vp8_filter_params (host) -> loop filter -> BGRA against
    the reference's whole-file decode of a WebP whose loop filter is on"""
    import ctypes as C
    from ffpic_amd import capi
    from test_oracle_golden import vp8_filter_header
    g = golden("webp_file_lf.npz")
    w, h, pitch = [int(x) for x in g[f"{tag}_dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    filt = np.zeros((4, 2, 3), np.uint8)
    ft = C.c_int(-1)
    hdr = vp8_filter_header(g[f"{tag}_lf"], g[f"{tag}_lf_header"])
    capi.check(capi.lib().ffhip_vp8_filter_params(C.byref(hdr), filt.ctypes.data, C.byref(ft)))
    y, u, v = ops.vp8_predict_recon(c, r, g[f"{tag}_modes"][None], g[f"{tag}_residual"][None])
    y, u, v = ops.vp8_loopfilter(c, r, ft.value, g[f"{tag}_modes"][None], filt, y, u, v)
    bgra = ops.yuv420_to_bgra(y, u, v, r, c, pitch=pitch)
    assert np.array_equal(bgra[0][:h], g[f"{tag}_bgra"])


def test_webp_file_1080p_real_encoder(golden):
    """BASELINE config 4 at its own size from a real encoder's stream (libwebp on a photograph mosaic, loop filter on,
    54 % B_PRED macroblocks): residual dump -> predict + reconstruct -> filter parameters (host) -> loop filter -> BGRA,
    every row against the reference's whole-file decode (per-row checksums, the first 32 rows byte by byte)"""
    import ctypes as C
    from ffpic_amd import capi
    from test_oracle_golden import vp8_filter_header
    g = golden("webp_file_1080p.npz")
    w, h, pitch = [int(x) for x in g["dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    assert (c, r) == (120, 68)
    filt = np.zeros((4, 2, 3), np.uint8)
    ft = C.c_int(-1)
    hdr = vp8_filter_header(g["lf"], g["lf_header"])
    capi.check(capi.lib().ffhip_vp8_filter_params(C.byref(hdr), filt.ctypes.data, C.byref(ft)))
    y, u, v = ops.vp8_predict_recon(c, r, g["modes"][None], g["residual"][None])
    y, u, v = ops.vp8_loopfilter(c, r, ft.value, g["modes"][None], filt, y, u, v)
    bgra = ops.yuv420_to_bgra(y, u, v, r, c, pitch=pitch)[0][:h]
    assert np.array_equal(bgra[:32], g["bgra_head"])
    rows = np.ascontiguousarray(bgra).reshape(h, -1).view(np.uint32).astype(np.uint64)
    sums = (rows * (np.arange(rows.shape[1], dtype=np.uint64) + np.uint64(1))).sum(axis=1, dtype=np.uint64)
    bad = np.nonzero(sums != g["bgra_row_sums"])[0]
    assert bad.size == 0, f"{bad.size} rows differ, first {bad[:5]}"


@pytest.mark.parametrize("ft", [1, 2])
def test_predict_and_loopfilter_side_by_side(ft):
    """ffhip_vp8_predict_loopfilter (both row kernels enqueued next to each other, the filter following the prediction through
    its progress counters) against the two calls one after the other and against the oracle: several 1080p frames, every byte"""
    c, r, n = 120, 68, 3
    modes = np.stack([synth.vp8_modes(c, r, seed=700 + i) for i in range(n)])
    modes[..., 18] = np.random.default_rng(11).integers(0, 4, size=modes[..., 18].shape)
    resid = np.stack([synth.vp8_residual(c * r, seed=710 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=13)
    y0, u0, v0 = ops.vp8_predict_recon(c, r, modes, resid)
    seq = ops.vp8_loopfilter(c, r, ft, modes, flt, y0, u0, v0)
    fused = ops.vp8_predict_loopfilter(c, r, modes, resid, ft, flt)
    for a, b, name in zip(seq, fused, "YUV"):
        assert np.array_equal(a, b), (ft, name)
    exp = oracle_lf(c, r, ft, modes[0], flt, (y0[0], u0[0], v0[0]))
    for gp, e, name in zip(fused, exp, "YUV"):
        assert np.array_equal(gp[0], e), (ft, name)


def test_predict_and_loopfilter_side_by_side_small_and_odd():
    """the side-by-side call on geometries with one row, one column, and a mode mix full of the wrapped H_PRED"""
    for (c, r, n, seed) in ((1, 1, 2, 1), (7, 1, 1, 2), (1, 9, 1, 3), (21, 13, 4, 4)):
        modes = np.stack([synth.vp8_modes(c, r, seed=720 + seed + i) for i in range(n)])
        modes[:, ::c, 0] = 3                                 # H_PRED in column 0 of every row: the row-to-row chain
        resid = np.stack([synth.vp8_residual(c * r, seed=730 + seed + i) for i in range(n)])
        flt = synth.vp8_filters(seed=15)
        for ft in (1, 2):
            y0, u0, v0 = ops.vp8_predict_recon(c, r, modes, resid)
            seq = ops.vp8_loopfilter(c, r, ft, modes, flt, y0, u0, v0)
            fused = ops.vp8_predict_loopfilter(c, r, modes, resid, ft, flt)
            for a, b, name in zip(seq, fused, "YUV"):
                assert np.array_equal(a, b), (c, r, n, ft, name)


def test_webp_file_1080p_side_by_side(golden):
    """the real encoder's 1080p frame through ffhip_vp8_predict_loopfilter: the reference's whole-file decode, every row"""
    import ctypes as C
    from ffpic_amd import capi
    from test_oracle_golden import vp8_filter_header
    g = golden("webp_file_1080p.npz")
    w, h, pitch = [int(x) for x in g["dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    filt = np.zeros((4, 2, 3), np.uint8)
    ft = C.c_int(-1)
    capi.check(capi.lib().ffhip_vp8_filter_params(C.byref(vp8_filter_header(g["lf"], g["lf_header"])), filt.ctypes.data, C.byref(ft)))
    modes = np.stack([g["modes"]] * 4)
    resid = np.stack([g["residual"]] * 4)
    y, u, v = ops.vp8_predict_loopfilter(c, r, modes, resid, ft.value, filt)
    for i in (0, 3):
        bgra = ops.yuv420_to_bgra(y[i:i + 1], u[i:i + 1], v[i:i + 1], r, c, pitch=pitch)[0][:h]
        assert np.array_equal(bgra[:32], g["bgra_head"])
        rows = np.ascontiguousarray(bgra).reshape(h, -1).view(np.uint32).astype(np.uint64)
        sums = (rows * (np.arange(rows.shape[1], dtype=np.uint64) + np.uint64(1))).sum(axis=1, dtype=np.uint64)
        assert np.array_equal(sums, g["bgra_row_sums"]), i


@pytest.mark.parametrize("entry", ["predict", "predict_loopfilter"])
def test_bad_mode_bytes_in_a_large_batch_are_refused_through_the_stream(entry):
    """Up to 2^17 macroblocks the mode bytes are checked on the host and the call returns FFHIP_EINVAL; beyond that a kernel in front of
    the row kernel checks them, the call itself returns 0, nothing is written and ffhip_stream_sync reports FFHIP_EINVAL -- once: the
    next sync is clean, and a valid call afterwards works (the side-by-side filter must neither hang nor report its own time-out)."""
    from ffpic_amd import capi
    L = capi.require_device()
    c, r, n = 120, 68, 17                                     # 138 720 macroblocks
    n_mb = c * r
    m0, r0 = synth.vp8_modes(c, r, seed=31), synth.vp8_residual(n_mb, seed=32)
    modes = np.ascontiguousarray(np.broadcast_to(m0, (n,) + m0.shape)).copy()
    resid = np.ascontiguousarray(np.broadcast_to(r0, (n,) + r0.shape))
    flt = synth.vp8_filters(seed=15)
    dm_good = ops.DeviceBuffer(modes)
    modes_bad = modes.copy()
    modes_bad[n - 1, n_mb - 3, 0] = 9                         # one record of the last frame: not a VP8 16x16 mode
    dm, dr, df = ops.DeviceBuffer(modes_bad), ops.DeviceBuffer(resid), ops.DeviceBuffer(np.ascontiguousarray(flt))
    ysz, csz = 256 * n_mb, 64 * n_mb
    dy, du, dv = ops.DeviceBuffer(nbytes=n * ysz), ops.DeviceBuffer(nbytes=n * csz), ops.DeviceBuffer(nbytes=n * csz)

    def call(host_modes, dev_modes):
        if entry == "predict":
            return L.ffhip_vp8_predict_recon(c, r, n, host_modes.ctypes.data, dev_modes.ptr, dr.ptr, n_mb * 384, None, dy.ptr, du.ptr, dv.ptr, ysz, csz, None)
        return L.ffhip_vp8_predict_loopfilter(c, r, n, host_modes.ctypes.data, dev_modes.ptr, dr.ptr, n_mb * 384, None, 2, df.ptr, dy.ptr, du.ptr, dv.ptr, ysz, csz, None)
    for d in (dy, du, dv):
        capi.check(L.ffhip_memset(d.ptr, 0x5a, d.nbytes, None))
    capi.check(L.ffhip_stream_sync(None))
    assert call(modes_bad, dm) == 0
    assert L.ffhip_stream_sync(None) == capi.FFHIP_EINVAL
    assert L.ffhip_stream_sync(None) == 0
    assert (dy.to_host((n * ysz,), np.uint8) == 0x5a).all() and (du.to_host((n * csz,), np.uint8) == 0x5a).all()
    assert call(modes, dm_good) == 0
    assert L.ffhip_stream_sync(None) == 0
    y = dy.to_host((n, 16 * r, 16 * c), np.uint8)
    assert np.array_equal(y[0], y[n - 1]) and not (y[0] == 0x5a).all()


@pytest.mark.parametrize("large", [False, True])
def test_sub_block_mode_above_9_is_refused(large):
    """A B_PRED record whose 4x4 mode byte is above 9 indexes past the reference's table of ten predictors (predict.c: PredLuma4):
    refused like a bad 16x16 mode -- by the call itself for small batches, through the stream for large ones; the same byte in a record
    that is NOT B_PRED is never looked at (the reference does not look at it either)."""
    from ffpic_amd import capi
    L = capi.require_device()
    c, r = 120, 68
    n = 17 if large else 1
    n_mb = c * r
    m0, r0 = synth.vp8_modes(c, r, seed=41), synth.vp8_residual(n_mb, seed=42)
    modes = np.ascontiguousarray(np.broadcast_to(m0, (n,) + m0.shape)).copy()
    resid = np.ascontiguousarray(np.broadcast_to(r0, (n,) + r0.shape))
    bp = np.flatnonzero(modes[n - 1, :, 0] == 4)
    nb = np.flatnonzero(modes[n - 1, :, 0] != 4)
    assert len(bp) and len(nb)
    harmless = modes.copy()
    harmless[n - 1, nb[-1], 2 + 7] = 200                     # not a B_PRED record: its imodes are not read
    bad = modes.copy()
    bad[n - 1, bp[-1], 2 + 15] = 10                          # the last sub-block of a B_PRED record
    dr = ops.DeviceBuffer(resid)
    ysz, csz = 256 * n_mb, 64 * n_mb
    dy, du, dv = ops.DeviceBuffer(nbytes=n * ysz), ops.DeviceBuffer(nbytes=n * csz), ops.DeviceBuffer(nbytes=n * csz)

    def call(m):
        dm = ops.DeviceBuffer(m)
        for d in (dy, du, dv):                                # the planes' initial contents matter (raw H_PRED reads at x = 0)
            capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
        rc = L.ffhip_vp8_predict_recon(c, r, n, m.ctypes.data, dm.ptr, dr.ptr, n_mb * 384, None, dy.ptr, du.ptr, dv.ptr, ysz, csz, None)
        rs = L.ffhip_stream_sync(None)
        return rc, rs
    assert call(bad) == ((0, capi.FFHIP_EINVAL) if large else (capi.FFHIP_EINVAL, 0))
    assert call(harmless) == (0, 0)
    y_h = dy.to_host((n, 16 * r, 16 * c), np.uint8)
    assert call(modes) == (0, 0)
    assert np.array_equal(dy.to_host((n, 16 * r, 16 * c), np.uint8), y_h)


def test_webp_file_1080p_256_frames(golden):
    """A chip-filling batch: 256 copies of the real encoder's 1080p frame in ONE ffhip_vp8_predict_loopfilter call (the frame loop
    of webp.c:1833-1866; the wave caps follow residency at this size, not 16 waves per image).  Every frame's planes equal the
    first frame's, and the first, the middle and the last frame are the reference's whole-file decode, every row."""
    import ctypes as C
    from ffpic_amd import capi
    from test_oracle_golden import vp8_filter_header
    g = golden("webp_file_1080p.npz")
    w, h, pitch = [int(x) for x in g["dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    filt = np.zeros((4, 2, 3), np.uint8)
    ft = C.c_int(-1)
    capi.check(capi.lib().ffhip_vp8_filter_params(C.byref(vp8_filter_header(g["lf"], g["lf_header"])), filt.ctypes.data, C.byref(ft)))
    n = 256
    modes = np.broadcast_to(g["modes"], (n,) + g["modes"].shape)
    resid = np.broadcast_to(g["residual"], (n,) + g["residual"].shape)
    y, u, v = ops.vp8_predict_loopfilter(c, r, np.ascontiguousarray(modes), np.ascontiguousarray(resid), ft.value, filt)
    for i in range(1, n):
        assert np.array_equal(y[i], y[0]) and np.array_equal(u[i], u[0]) and np.array_equal(v[i], v[0]), i
    for i in (0, n // 2, n - 1):
        bgra = ops.yuv420_to_bgra(y[i:i + 1], u[i:i + 1], v[i:i + 1], r, c, pitch=pitch)[0][:h]
        assert np.array_equal(bgra[:32], g["bgra_head"])
        rows = np.ascontiguousarray(bgra).reshape(h, -1).view(np.uint32).astype(np.uint64)
        sums = (rows * (np.arange(rows.shape[1], dtype=np.uint64) + np.uint64(1))).sum(axis=1, dtype=np.uint64)
        assert np.array_equal(sums, g["bgra_row_sums"]), i


@pytest.mark.parametrize("env", [{"FFHIP_VP8_PRED_MODE": "levels"}, {"FFHIP_VP8_LF_MODE": "levels"}, {"FFHIP_VP8_FUSE": "0"},
                                 {"FFHIP_VP8_PRED_WAVES": "5", "FFHIP_VP8_LF_WAVES": "3"},
                                 {"FFHIP_VP8_PRED_SPLIT": "0"}, {"FFHIP_VP8_PRED_SPLIT": "1"},
                                 {"FFHIP_VP8_PRED_SPLIT": "1", "FFHIP_VP8_PRED_WAVES": "2", "FFHIP_VP8_LF_WAVES": "3"},
                                 {"FFHIP_VP8_PRED_SPLIT": "1", "FFHIP_VP8_PRED_WAVES": "1", "FFHIP_VP8_FUSE": "0"},
                                 {"FFHIP_VP8_PROGRESS_SHIFT": "0"}, {"FFHIP_VP8_PROGRESS_SHIFT": "3", "FFHIP_VP8_PRED_SPLIT": "1"}, {"FFHIP_SIDE_PRIORITY": "0"}])
def test_side_by_side_call_under_every_scheduler(env, monkeypatch):
    """ffhip_vp8_predict_loopfilter when one of its stages cannot take the row form (then they run one after the other), when
    told not to overlap, with far fewer waves than rows (tickets, not residency, order the rows of both kernels), and in both
    forms of the prediction's rows -- one wave per macroblock row, or luma rows and chroma rows on different waves with counters
    of their own that the filter waits for too (with two waves: one of each kind; with one, not overlapped: it does both)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v); capi.reload_env()
    c, r, n = 21, 13, 3
    modes = np.stack([synth.vp8_modes(c, r, seed=760 + i) for i in range(n)])
    resid = np.stack([synth.vp8_residual(c * r, seed=770 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=17)
    got = ops.vp8_predict_loopfilter(c, r, modes, resid, 2, flt)
    for i in range(n):
        y0, u0, v0 = O.oracle_vp8_frame(c, r, modes[i], resid[i])
        exp = oracle_lf(c, r, 2, modes[i], flt, (y0, u0, v0))
        for gp, e, name in zip(got, exp, "YUV"):
            assert np.array_equal(gp[i], e), (env, i, name)


@pytest.mark.parametrize("env", [{"FFHIP_VP8_LF_MODE": "levels"}, {"FFHIP_VP8_LF_WAVES": "3"}, {}])
@pytest.mark.parametrize("ft", [1, 2])
def test_lf_schedulers_agree(env, ft, monkeypatch):
    """level-synchronous launches vs the single row-form launch (few / many waves), both filter types"""
    for k, v in env.items():
        monkeypatch.setenv(k, v); capi.reload_env()
    c, r, n = 21, 13, 3
    modes = np.stack([synth.vp8_modes(c, r, seed=500 + i) for i in range(n)])
    modes[..., 18] = np.random.default_rng(5).integers(0, 4, size=modes[..., 18].shape)
    flt = synth.vp8_filters(seed=7)
    planes = [synth.vp8_blocky_planes(c, r, seed=510 + i) for i in range(n)]
    y = np.stack([p[0] for p in planes]); u = np.stack([p[1] for p in planes]); v = np.stack([p[2] for p in planes])
    got = ops.vp8_loopfilter(c, r, ft, modes, flt, y, u, v)
    for i in range(n):
        exp = oracle_lf(c, r, ft, modes[i], flt, (y[i], u[i], v[i]))
        for gp, e, name in zip(got, exp, "YUV"):
            assert np.array_equal(gp[i], e), (env, ft, i, name)


def test_lf_row_handoff_stress():
    """several 1080p frames at once, every byte against the oracle"""
    c, r, n = 120, 68, 4
    modes = np.stack([synth.vp8_modes(c, r, seed=600 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=9)
    planes = [synth.vp8_blocky_planes(c, r, seed=610 + i) for i in range(n)]
    y = np.stack([p[0] for p in planes]); u = np.stack([p[1] for p in planes]); v = np.stack([p[2] for p in planes])
    for ft in (1, 2):
        got = ops.vp8_loopfilter(c, r, ft, modes, flt, y, u, v)
        for i in range(n):
            exp = oracle_lf(c, r, ft, modes[i], flt, (y[i], u[i], v[i]))
            for gp, e, name in zip(got, exp, "YUV"):
                assert np.array_equal(gp[i], e), (ft, i, name)
