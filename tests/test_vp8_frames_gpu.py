"""GPU: ffhip_vp8_decode_frames -- the frame loop of vp8_decode (format/webp.c:1833-1868: prediction, loop filter, colour
conversion) as one call, in its fused form (one workgroup per frame, one wave per macroblock row, every pixel stored once) and
in its three-stage form, against the oracle chain and against the reference's whole-file decodes."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth
from test_vp8_lf_gpu import oracle_lf

pytestmark = pytest.mark.gpu


def oracle_chain(c, r, ft, modes, resid, filt, resmap=None):
    """prediction -> loop filter -> BGRA of one frame by the CPU restatement; returns (bgra [16r][64c], planes)"""
    planes = O.oracle_vp8_frame(c, r, modes, resid, resmap)
    if ft:
        planes = oracle_lf(c, r, ft, modes, filt, planes)
    out = np.zeros((16 * r, 16 * c * 4), np.uint8)
    O.ffo().ffo_yuv420_to_bgra32(out.reshape(-1), 16 * c * 4, np.ascontiguousarray(planes[0]).reshape(-1), np.ascontiguousarray(planes[1]).reshape(-1),
                                 np.ascontiguousarray(planes[2]).reshape(-1), 16 * c, 8 * c, r, c)
    return out, planes


def force(monkeypatch, form, waves=None, grid=None):
    monkeypatch.setenv("FFHIP_VP8_FRAMES", form)
    if waves:
        monkeypatch.setenv("FFHIP_VP8_FRAME_WAVES", str(waves))
    if grid:
        monkeypatch.setenv("FFHIP_VP8_FRAME_GRID", str(grid))
    capi.reload_env()


@pytest.mark.parametrize("c,r,n,ft,waves", [(1, 1, 1, 2, 1), (1, 1, 3, 1, 4), (2, 1, 2, 2, 2), (1, 3, 2, 2, 2), (3, 2, 1, 0, 16), (5, 4, 3, 2, 3),
                                            (17, 9, 2, 1, 8), (17, 9, 5, 2, 16), (21, 13, 3, 2, 1), (21, 13, 7, 2, 5)])
def test_fused_vs_oracle_small_and_odd(c, r, n, ft, waves, monkeypatch):
    """every edge of the scheme on small pictures: one macroblock, one row, one column, more waves than rows, one wave for all
    rows, frames shared out over fewer workgroups than frames; BGRA and the optional planes against the oracle chain"""
    force(monkeypatch, "fused", waves, grid=2 if n > 2 else None)
    rng = np.random.default_rng(c * 100 + r * 10 + n)
    modes = np.stack([synth.vp8_modes(c, r, seed=900 + i) for i in range(n)])
    modes[..., 18] = rng.integers(0, 4, size=modes[..., 18].shape)
    resid = np.stack([synth.vp8_residual(c * r, seed=910 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=c + r)
    bgra, planes = ops.vp8_decode_frames(c, r, modes, resid, ft, flt, planes=True)
    bgra2 = ops.vp8_decode_frames(c, r, modes, resid, ft, flt)          # without the planes: the same pixels
    assert np.array_equal(bgra, bgra2)
    for i in range(n):
        exp, ep = oracle_chain(c, r, ft, modes[i], resid[i], flt)
        for gp, e, name in zip(planes, ep, "YUV"):
            assert np.array_equal(gp[i], e), (i, name, np.argwhere(gp[i] != e)[:4])
        assert np.array_equal(bgra[i], exp), (i, np.argwhere(bgra[i] != exp)[:4])


def test_fused_raw_h_pred_in_the_first_column(monkeypatch):
    """16x16 H_PRED at x = 0 reads the last pixel of the row above and, below it, samples not reconstructed yet (0 in the fresh
    planes of vp8_decode): a row that must wait for the WHOLE row above; V_PRED in the first row reads the bytes before the plane"""
    force(monkeypatch, "fused", 4)
    c, r, n = 9, 7, 2
    modes = np.stack([synth.vp8_modes(c, r, seed=930 + i) for i in range(n)])
    m = modes.reshape(n, r, c, 20)
    m[:, 1:, 0, 0] = 3          # every row but the first starts with H_PRED
    m[:, 0, ::2, 0] = 2         # V_PRED along the first row
    m[:, 0, 0, 0] = 3           # and H_PRED in the corner
    resid = np.stack([synth.vp8_residual(c * r, seed=940 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=3)
    bgra = ops.vp8_decode_frames(c, r, modes, resid, 2, flt)
    for i in range(n):
        exp, _ = oracle_chain(c, r, 2, modes[i], resid[i], flt)
        assert np.array_equal(bgra[i], exp), i


def test_fused_with_residual_map_and_pitch(monkeypatch):
    """skipped macroblocks keep the previous macroblock's coefficients (webp.c:1207-1223: a residual map); a BGRA pitch wider than the picture"""
    force(monkeypatch, "fused", 8)
    c, r, n = 12, 6, 3
    n_mb = c * r
    rng = np.random.default_rng(8)
    modes = np.stack([synth.vp8_modes(c, r, seed=950 + i) for i in range(n)])
    resid = np.stack([synth.vp8_residual(n_mb, seed=960 + i) for i in range(n)])
    resmap = np.stack([np.maximum.accumulate(np.where(rng.random(n_mb) < 0.3, 0, np.arange(n_mb))) for _ in range(n)]).astype(np.int32)
    flt = synth.vp8_filters(seed=5)
    pitch = 16 * c * 4 + 64
    bgra = ops.vp8_decode_frames(c, r, modes, resid, 2, flt, resmap=resmap, pitch=pitch)
    for i in range(n):
        exp, _ = oracle_chain(c, r, 2, modes[i], resid[i], flt, resmap[i])
        assert np.array_equal(bgra[i][:, :16 * c * 4], exp), i
        assert not bgra[i][:, 16 * c * 4:].any()


@pytest.mark.parametrize("form", ["fused", "rows", "auto"])
def test_both_forms_on_1080p_frames(form, monkeypatch):
    """a handful of 1080p frames of random modes: the fused kernel, the three stages, and whatever the entry picks by itself"""
    if form != "auto":
        force(monkeypatch, form)
    c, r, n = 120, 68, 3
    modes = np.stack([synth.vp8_modes(c, r, seed=970 + i) for i in range(n)])
    resid = np.stack([synth.vp8_residual(c * r, seed=980 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=11)
    bgra = ops.vp8_decode_frames(c, r, modes, resid, 2, flt)
    for i in range(n):
        exp, _ = oracle_chain(c, r, 2, modes[i], resid[i], flt)
        assert np.array_equal(bgra[i], exp), (form, i)


def _file_1080p(golden):
    from test_oracle_golden import vp8_filter_header
    g = golden("webp_file_1080p.npz")
    w, h, pitch = [int(x) for x in g["dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    filt = np.zeros((4, 2, 3), np.uint8)
    ft = C.c_int(-1)
    capi.check(capi.lib().ffhip_vp8_filter_params(C.byref(vp8_filter_header(g["lf"], g["lf_header"])), filt.ctypes.data, C.byref(ft)))
    return g, c, r, h, pitch, filt, ft.value


def _is_reference_decode(g, bgra, h):
    assert np.array_equal(bgra[:32], g["bgra_head"])
    rows = np.ascontiguousarray(bgra[:h]).reshape(h, -1).view(np.uint32).astype(np.uint64)
    sums = (rows * (np.arange(rows.shape[1], dtype=np.uint64) + np.uint64(1))).sum(axis=1, dtype=np.uint64)
    bad = np.nonzero(sums != g["bgra_row_sums"])[0]
    assert bad.size == 0, f"{bad.size} rows differ, first {bad[:5]}"


@pytest.mark.parametrize("n,waves", [(1, 16), (3, 8), (256, None)])
def test_fused_is_the_reference_whole_file_decode_1080p(golden, n, waves, monkeypatch):
    """BASELINE config 4 from a real encoder's stream (libwebp, 54 % B_PRED macroblocks, loop filter on): the fused kernel's BGRA of
    every frame equals the first one's, and the first, middle and last are the reference's whole-file decode, every row"""
    force(monkeypatch, "fused", waves)
    g, c, r, h, pitch, filt, ft = _file_1080p(golden)
    modes = np.ascontiguousarray(np.broadcast_to(g["modes"], (n,) + g["modes"].shape))
    resid = np.ascontiguousarray(np.broadcast_to(g["residual"], (n,) + g["residual"].shape))
    bgra = ops.vp8_decode_frames(c, r, modes, resid, ft, filt, pitch=pitch, host_modes=n < 16)
    for i in range(1, n):
        assert np.array_equal(bgra[i], bgra[0]), i
    for i in sorted({0, n // 2, n - 1}):
        _is_reference_decode(g, bgra[i], h)


@pytest.mark.parametrize("tag", ["q55", "q40"])
def test_fused_small_webp_files_with_loop_filter(golden, tag, monkeypatch):
    """the two small whole-file fixtures whose loop filter is on, and the one without (level 0)"""
    from test_oracle_golden import vp8_filter_header
    force(monkeypatch, "fused", 4)
    g = golden("webp_file_lf.npz")
    w, h, pitch = [int(x) for x in g[f"{tag}_dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    filt = np.zeros((4, 2, 3), np.uint8)
    ft = C.c_int(-1)
    capi.check(capi.lib().ffhip_vp8_filter_params(C.byref(vp8_filter_header(g[f"{tag}_lf"], g[f"{tag}_lf_header"])), filt.ctypes.data, C.byref(ft)))
    bgra = ops.vp8_decode_frames(c, r, g[f"{tag}_modes"][None], g[f"{tag}_residual"][None], ft.value, filt, pitch=pitch)
    assert np.array_equal(bgra[0][:h], g[f"{tag}_bgra"])
    g0 = golden("webp_file.npz")
    w, h, pitch = [int(x) for x in g0["dims"]]
    c, r = (w + 15) // 16, (h + 15) // 16
    bgra = ops.vp8_decode_frames(c, r, g0["modes"][None], g0["residual"][None], 0, None, pitch=pitch)
    assert np.array_equal(bgra[0][:h], g0["bgra"])


def test_fused_refuses_bad_modes_through_the_stream(monkeypatch):
    """a large batch is checked by the kernel in front: the call returns 0, nothing is written, the next sync says FFHIP_EINVAL once"""
    force(monkeypatch, "fused")
    L = capi.require_device()
    c, r, n = 120, 68, 17
    n_mb = c * r
    m0, r0 = synth.vp8_modes(c, r, seed=31), synth.vp8_residual(n_mb, seed=32)
    modes = np.ascontiguousarray(np.broadcast_to(m0, (n,) + m0.shape)).copy()
    resid = np.ascontiguousarray(np.broadcast_to(r0, (n,) + r0.shape))
    bad = modes.copy()
    bad[n - 1, n_mb - 3, 0] = 9
    flt = np.ascontiguousarray(synth.vp8_filters(seed=15))
    dr, df = ops.DeviceBuffer(resid), ops.DeviceBuffer(flt)
    H, W = 16 * r, 16 * c
    do = ops.DeviceBuffer(nbytes=n * H * W * 4)

    def call(m, host):
        dm = ops.DeviceBuffer(m)
        capi.check(L.ffhip_memset(do.ptr, 0x5a, do.nbytes, None))
        rc = L.ffhip_vp8_decode_frames(c, r, n, m.ctypes.data if host else None, dm.ptr, dr.ptr, n_mb * 384, None, 2, df.ptr, do.ptr, W * 4, H * W * 4,
                                       None, None, None, 0, 0, None)
        return rc, L.ffhip_stream_sync(None)
    assert call(bad, True) == (0, capi.FFHIP_EINVAL)
    assert (do.to_host((n * H * W * 4,), np.uint8) == 0x5a).all()
    assert L.ffhip_stream_sync(None) == 0
    assert call(modes, False) == (0, 0)
    out = do.to_host((n, H, W * 4), np.uint8)
    assert np.array_equal(out[0], out[n - 1]) and not (out[0] == 0x5a).all()
    exp, _ = oracle_chain(c, r, 2, modes[0], resid[0], flt)
    assert np.array_equal(out[n - 1], exp)


def test_fused_handoff_stress(monkeypatch):
    """many frames of different content per workgroup, few waves per frame and more workgroups than the chip holds at once: the
    line buffers are reused by every row and every frame of a workgroup's share; every byte of every frame against the oracle"""
    force(monkeypatch, "fused", 3, grid=5)
    c, r, n = 40, 23, 23
    modes = np.stack([synth.vp8_modes(c, r, seed=1000 + i) for i in range(n)])
    modes[..., 18] = np.random.default_rng(4).integers(0, 4, size=modes[..., 18].shape)
    resid = np.stack([synth.vp8_residual(c * r, seed=1100 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=21)
    for _ in range(3):
        bgra = ops.vp8_decode_frames(c, r, modes, resid, 2, flt)
        for i in range(n):
            exp, _ = oracle_chain(c, r, 2, modes[i], resid[i], flt)
            assert np.array_equal(bgra[i], exp), i
