"""GPU: HEVC intra prediction + reconstruction against goldens and the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth

pytestmark = pytest.mark.gpu


def assert_wavefront_schedule(sorted_by_plane=None):
    """what the device planner made of the list of the call just made: taken (not left to the one-wave serial kernel), tickets by coding-tree
    wavefront (not decode order), 64x64 luma windows; sorted_by_plane: whether the list had to be sorted by plane for that"""
    import ctypes as C
    out = (C.c_uint32 * 8)()
    capi.check(capi.lib().ffhip_debug_hevc_plan_result(out), "ffhip_debug_hevc_plan_result")
    assert out[6] == 0 and out[0] == 0 and out[1] > 0, list(out)
    assert out[3] == 0 and out[4] > 0, list(out)
    assert out[5] == 6, list(out)
    if sorted_by_plane is not None:
        assert bool(out[7]) == sorted_by_plane, list(out)
    return list(out)


def test_golden_tu_lists(golden):
    g = golden("hevc_intra.npz")
    for tag in "abcde":     # d, e: 4:4:4 with cross-component prediction
        w, h, bd, bdc, csub = [int(x) for x in g[f"{tag}_dims"]]
        tus = np.ascontiguousarray(g[f"{tag}_tus"]).view(synth.HEVC_TU_DTYPE).reshape(-1)
        y, u, v = ops.hevc_intra_recon(tus, g[f"{tag}_residual"], w, h, True, bd, bdc, csub=csub)
        assert np.array_equal(y, g[f"{tag}_y"]), tag
        assert np.array_equal(u, g[f"{tag}_u"]) and np.array_equal(v, g[f"{tag}_v"]), tag


@pytest.mark.parametrize("w,h,seed,adv,bd", [(64, 64, 1, False, 8), (192, 128, 2, True, 8), (256, 256, 3, False, 10),
                                             (320, 192, 4, True, 12), (512, 256, 5, False, 8)])
def test_tu_lists_vs_oracle(w, h, seed, adv, bd):
    tus, res = synth.hevc_intra_tus(w, h, seed + 100, adversarial_masks=adv)
    if bd > 8:
        res = (res.astype(np.int32) * (1 << (bd - 8))).astype(np.int16)
    got = ops.hevc_intra_recon(tus, res, w, h, True, bd, bd)
    exp = O.oracle_hevc_intra(tus, res, w, h, True, bd, bd)
    for gp, e, name in zip(got, exp, "YUV"):
        assert np.array_equal(gp, e), name


@pytest.mark.parametrize("w,h,seed,bd,bdc", [(128, 128, 1, 8, 8), (256, 192, 2, 10, 12), (192, 64, 3, 12, 8)])
def test_cross_component_444_vs_oracle(w, h, seed, bd, bdc):
    """8.6.6 after rdpcm on 4:4:4 TU lists, with extreme residuals so the shift/multiply wraps"""
    tus, res = synth.hevc_intra_tus(w, h, seed + 300, adversarial_masks=True, ccp=True, chroma_444=True)
    res = res.copy()
    res[::53] = 32767
    res[7::59] = -32768
    assert (tus["flags"] & synth.TU_CCP).sum() > 20
    got = ops.hevc_intra_recon(tus, res, w, h, True, bd, bdc, csub=1)
    exp = O.oracle_hevc_intra(tus, res, w, h, True, bd, bdc, csub=1)
    for gp, e, name in zip(got, exp, "YUV"):
        assert np.array_equal(gp, e), name


def test_every_mode_and_size_isolated():
    """each (mode, size) alone in the middle of a noisy picture, with and without smoothing"""
    rng = np.random.default_rng(0)
    w = h = 128
    for lg in (2, 3, 4, 5):
        n = 1 << lg
        recs, parts, off = [], [], 0
        # a first TU that fills the surroundings cannot exist in one list; instead use availability
        # patterns: nothing / corner+top / left only / everything that lies inside the picture
        for mode in range(35):
            for fl_extra in (0, synth.TU_FILTER | synth.TU_STRONG):
                x0, y0 = 32, 32
                recs.append((x0, y0, lg, 0, mode, fl_extra | synth.TU_RESIDUAL, off, 0, 0, 0))
                parts.append(rng.integers(-40, 300, size=n * n).astype(np.int16))
                off += n * n
        tus = np.array(recs, dtype=synth.HEVC_TU_DTYPE)
        res = np.concatenate(parts)
        # each TU overwrites the same area; they are independent (no available neighbours) so the
        # library may run them in any order: run them one at a time to keep the result defined
        for i in range(len(tus)):
            got = ops.hevc_intra_recon(tus[i:i + 1], res, w, h, False)[0]
            exp = O.oracle_hevc_intra(tus[i:i + 1], res, w, h, False)[0]
            assert np.array_equal(got, exp), (lg, i)


@pytest.mark.parametrize("env", [{"FFHIP_HEVC_INTRA_MODE": "levels"}, {"FFHIP_HEVC_PLAN": "host"}, {"FFHIP_HEVC_PLAN": "host", "FFHIP_HEVC_INTRA_WINDOW": "4"},
                                 {"FFHIP_HEVC_INTRA_WINDOW": "3"},
                                 {"FFHIP_HEVC_INTRA_WINDOW": "4"}, {"FFHIP_HEVC_INTRA_WINDOW": "5"},
                                 {"FFHIP_HEVC_INTRA_WINDOW": "6"}, {"FFHIP_HEVC_INTRA_WAVES": "3"},
                                 {"FFHIP_HEVC_DEPTH_DIAGONALS": "1"}, {"FFHIP_HEVC_TICKET_SHARDS": "1"}, {"FFHIP_HEVC_INTRA_WIDTH_PCT": "1"},
                                 {"FFHIP_HEVC_BY_PLANE": "0"}, {"FFHIP_HEVC_BY_PLANE": "1"}, {"FFHIP_HEVC_BY_PLANE": "1", "FFHIP_HEVC_PLAN": "host"},
                                 {"FFHIP_HEVC_INTRA_DECODE_ORDER": "1", "FFHIP_HEVC_PLAN": "host"}, {"FFHIP_HEVC_INTRA_POLL_REPS": "4"},
                                 {"FFHIP_PLAN_THREADS": "3", "FFHIP_HEVC_PLAN": "host"}])
def test_schedulers_agree(env, monkeypatch):
    """the level-synchronous launches and the grouped single launch (any window) give the oracle's picture -- on lists with each coding
    tree block's planes one after the other and on the same lists in the reference's order (per coding unit: luma, Cb, Cr)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v); capi.reload_env()
    for (w, h, seed, adv, c444) in ((256, 192, 41, False, False), (192, 128, 42, True, False), (128, 128, 43, True, True)):
        tus, res = synth.hevc_intra_tus(w, h, seed, adversarial_masks=adv, ccp=c444, chroma_444=c444)
        csub = 1 if c444 else 2
        exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8, csub=csub)
        for order, lst in (("plane", tus), ("reference", synth.hevc_reference_order(tus, 64, csub, seed))):
            got = ops.hevc_intra_recon(lst, res, w, h, True, 8, 8, csub=csub)
            for gp, e, name in zip(got, exp, "YUV"):
                assert np.array_equal(gp, e), (env, order, name)


def test_grouped_form_small_ctb_falls_back_to_smaller_window():
    """a 32x32 coding tree block list with the default 64 window would make groups wait for later
    groups; the planner has to shrink the window (or fall back) and still be exact"""
    tus, res = synth.hevc_intra_tus(256, 128, 44, ctb=32)
    got = ops.hevc_intra_recon(tus, res, 256, 128, True, 8, 8)
    exp = O.oracle_hevc_intra(tus, res, 256, 128, True, 8, 8)
    for gp, e, name in zip(got, exp, "YUV"):
        assert np.array_equal(gp, e), name


@pytest.mark.parametrize("w,h,ctb,seed", [(80, 48, 16, 71), (96, 160, 32, 72), (208, 112, 16, 73)])
def test_pictures_that_are_not_a_whole_number_of_windows(w, h, ctb, seed):
    """picture sizes that are multiples of a small coding tree block only: the 64x64 scheduling windows, the per-pixel
    program words and the substitution table all hang over the picture's edge"""
    tus, res = synth.hevc_intra_tus(w, h, seed, ctb=ctb, adversarial_masks=True)
    got = ops.hevc_intra_recon(tus, res, w, h, True, 8, 8)
    exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8)
    for gp, e, name in zip(got, exp, "YUV"):
        assert np.array_equal(gp, e), name


@pytest.mark.parametrize("w,h,seed,bd", [(192, 128, 61, 8), (128, 192, 62, 10)])
def test_heic_chain_levels_to_bgra(w, h, seed, bd):
    """BASELINE config 5 in small: quantised levels -> ffhip_hevc_residual_batch per TU size -> ffhip_hevc_intra_recon
    -> ffhip_yuv420_to_bgra_16, against the same chain through the oracle (scale_and_transform, decode_intra_block,
    YUV420_to_BGRA32_16bit of the reference restated)"""
    from test_hevc_gpu import oracle_tus
    from test_color_gpu import oracle_420_16
    rng = np.random.default_rng(seed)
    tus, _ = synth.hevc_intra_tus(w, h, seed, adversarial_masks=False)
    tus = tus.copy()
    tus["flags"] &= ~np.uint8(synth.TU_RDPCM)                       # rdpcm needs transform-skip TUs: keep the chain plain
    res_gpu = np.zeros(int(sum(1 << (2 * int(t["log2_size"])) for t in tus)) + 16, np.int16)
    res_ora = np.zeros_like(res_gpu)
    off = 0
    order = {}
    for i, t in enumerate(tus):
        n = 1 << int(t["log2_size"])
        tus["res_offset"][i] = off
        order.setdefault(n, []).append((i, off))
        off += n * n
    for n, items in order.items():                                  # one residual batch per TU size, as a decoder would issue them
        lv = np.rint(rng.laplace(0, 6, size=(len(items), n * n))).astype(np.int16)
        info = np.zeros((len(items), 4), np.uint8)
        info[:, 0] = rng.integers(20, 38, size=len(items))          # qP
        for k, (i, _) in enumerate(items):
            if n == 4 and int(tus["cidx"][i]) == 0:
                info[k, 1] = 1                                      # luma intra 4x4: DST-VII (hevc.c:3911-3920)
        g = ops.hevc_residual_batch(n, lv, info, bitdepth=bd)
        o = oracle_tus(n, lv, info, bd, 0, None)
        for k, (_, o0) in enumerate(items):
            res_gpu[o0:o0 + n * n] = g[k]
            res_ora[o0:o0 + n * n] = o[k]
    assert np.array_equal(res_gpu, res_ora)
    got = ops.hevc_intra_recon(tus, res_gpu, w, h, True, bd, bd)
    exp = O.oracle_hevc_intra(tus, res_ora, w, h, True, bd, bd)
    for gp, e, name in zip(got, exp, "YUV"):
        assert np.array_equal(gp, e), name
    ctb = 64
    bgra = ops.yuv420_to_bgra_16(got[0][None], got[1][None], got[2][None], h // ctb, w // ctb, ctb)[0]
    assert np.array_equal(bgra, oracle_420_16(exp[0], exp[1], exp[2], h // ctb, w // ctb, ctb))


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])     # "e", "f": through the reference's HEIF loader, from tests/golden/file_e.heic (one image item) and file_f_grid.heic (a 1 x 1 grid item)
def test_hevc_file_config5(golden, tag):
    """f4 / BASELINE config 5 from the reference's own parse: quantised levels and TU lists its decoder recorded while
    decoding a hand-assembled HEVC stream -> ffhip_hevc_residual_batch per TU size -> ffhip_hevc_intra_recon ->
    ffhip_yuv420_to_bgra_16, against the residuals, planes and BGRA picture the reference itself produced"""
    g = golden("hevc_file.npz")
    w, h, _ = [int(x) for x in g[f"{tag}_dims"]]
    tus = np.ascontiguousarray(g[f"{tag}_tus"]).view(synth.HEVC_TU_DTYPE).reshape(-1).copy()
    info, lv = g[f"{tag}_tuinfo"], g[f"{tag}_levels"]
    has = (tus["flags"] & synth.TU_RESIDUAL) != 0
    resid = np.zeros(len(lv) + 16, np.int16)
    for lg in (2, 3, 4, 5):                                   # one residual batch per TU size, as a decoder would issue them
        idx = np.nonzero(has & (tus["log2_size"] == lg))[0]
        if idx.size == 0:
            continue
        n = 1 << lg
        offs = tus["res_offset"][idx].astype(np.int64)
        levels = np.stack([lv[o:o + n * n] for o in offs])
        got = ops.hevc_residual_batch(n, levels, np.ascontiguousarray(info[idx]), bitdepth=8)
        for k, o in enumerate(offs):
            resid[o:o + n * n] = got[k]
    assert np.array_equal(resid[:len(lv)], g[f"{tag}_resid"])
    y, u, v = ops.hevc_intra_recon(tus, resid, w, h, True, 8, 8)
    assert_wavefront_schedule()
    assert np.array_equal(y, g[f"{tag}_y"]) and np.array_equal(u, g[f"{tag}_u"]) and np.array_equal(v, g[f"{tag}_v"])
    bgra = ops.yuv420_to_bgra_16(y[None], u[None], v[None], h // 64, w // 64, 64)[0]
    assert np.array_equal(bgra, g[f"{tag}_bgra"])


def test_hevc_file_1080p(golden):
    """The same chain on the 1920x1080 stream (93 330 TUs; the bottom row of coding tree blocks is cut at 56 of 64 lines): inputs as
    the reference's parser recorded them, results against SHA-256 of the reference's own residuals, planes and BGRA rows
    (tests/golden/make_golden.py::gen_hevc_file_1080p).  The colour step is called the way hevc.c:7261-7263 calls it: 17 rows of
    coding tree blocks over ONE allocation holding Y, U at w*h and V at w*h*3/2, so that the rows past the picture read what the
    reference reads; the picture's 1080 rows are compared."""
    import hashlib

    def sha(a):
        return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)
    g = golden("hevc_file_1080p.npz")
    w, h = int(g["g_dims"][0]), int(g["g_dims"][1])
    tus = np.ascontiguousarray(g["g_tus"]).view(synth.HEVC_TU_DTYPE).reshape(-1).copy()
    info, lv = g["g_tuinfo"], g["g_levels"]
    has = (tus["flags"] & synth.TU_RESIDUAL) != 0
    resid = np.zeros(len(lv) + 16, np.int16)
    for lg in (2, 3, 4, 5):
        idx = np.nonzero(has & (tus["log2_size"] == lg))[0]
        n = 1 << lg
        offs = tus["res_offset"][idx].astype(np.int64)
        levels = lv[(offs[:, None] + np.arange(n * n)[None, :]).reshape(-1)].reshape(len(idx), n * n)
        got = ops.hevc_residual_batch(n, levels, np.ascontiguousarray(info[idx]), bitdepth=8)
        resid[(offs[:, None] + np.arange(n * n)[None, :]).reshape(-1)] = got.reshape(-1)
    assert np.array_equal(sha(resid[:len(lv)]), g["g_sha_resid"])
    y, u, v = ops.hevc_intra_recon(tus, resid, w, h, True, 8, 8)
    assert_wavefront_schedule(sorted_by_plane=True)    # 56 888 plane switches in 93 330 records
    assert np.array_equal(sha(y), g["g_sha_y"]) and np.array_equal(sha(u), g["g_sha_u"]) and np.array_equal(sha(v), g["g_sha_v"])
    size, rows = w * h, -(-h // 64) * 64
    planes = np.zeros(2 * size, np.int16)
    planes[:size], planes[size:size + size // 4], planes[size * 3 // 2:size * 3 // 2 + size // 4] = y.reshape(-1), u.reshape(-1), v.reshape(-1)
    L = capi.require_device()
    dp, do = ops.DeviceBuffer(planes), ops.DeviceBuffer(nbytes=rows * w * 4)
    capi.check(L.ffhip_yuv420_to_bgra_16(do.ptr, w * 4, dp.ptr, dp.ptr + 2 * size, dp.ptr + 3 * size, w, w // 2, rows // 64, w // 64, 64, 1, 0, 0, 0, None))
    bgra = do.to_host((rows, w * 4), np.uint8)
    assert np.array_equal(bgra[[0, h // 2, h - 1]], g["g_rows"])
    assert np.array_equal(sha(bgra[:h]), g["g_sha_bgra"])


@pytest.mark.parametrize("shift", [1, 3, 8, 13])
def test_residual_blocks_at_odd_offsets(shift):
    """Residual blocks need not be 16-byte aligned in d_residual: the kernel fetches aligned blocks ahead with 16-byte
    loads and reads the others where they are used.  Same list, every block moved by `shift` samples."""
    w, h = 192, 128
    tus, res = synth.hevc_intra_tus(w, h, seed=77)
    want = ops.hevc_intra_recon(tus, res, w, h, True, 8, 8)
    moved = tus.copy()
    moved["res_offset"] += shift
    res2 = np.concatenate([np.full(shift, 12345, np.int16), res])
    got = ops.hevc_intra_recon(moved, res2, w, h, True, 8, 8)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("plan", [None, "device"])
def test_list_the_device_planner_refuses_takes_the_serial_path(monkeypatch, plan):
    """4x4 TUs of a whole picture in RASTER order (not coding-tree order): every scheduling window is entered many times,
    so no window gives contiguous groups.  By default the host sees that from the list alone and takes its own planner (or
    the levels form); with FFHIP_HEVC_PLAN=device the list goes to the device planner all the same, which refuses it, and
    k_hevc_intra_serial decodes it with one wave in list order -- nobody on the host having looked at the verdict."""
    if plan:
        monkeypatch.setenv("FFHIP_HEVC_PLAN", plan); capi.reload_env()
    rng = np.random.default_rng(12)
    w, h = 96, 48
    recs, parts, off = [], [], 0
    done = np.zeros((h, w), bool)
    for y0 in range(0, h, 4):
        for x0 in range(0, w, 4):
            at = al = 0
            for k in range(8):
                if y0 > 0 and x0 + k < w and done[y0 - 1, x0 + k]:
                    at |= 1 << k
                if x0 > 0 and y0 + k < h and done[y0 + k, x0 - 1]:
                    al |= 1 << k
            fl = synth.TU_RESIDUAL | synth.TU_FILTER | (synth.TU_CORNER if x0 > 0 and y0 > 0 else 0)
            recs.append((x0, y0, 2, 0, int(rng.integers(0, 35)), fl, off, 0, at, al))
            parts.append(np.rint(rng.laplace(0, 12, size=16)).astype(np.int16))
            off += 16
            done[y0:y0 + 4, x0:x0 + 4] = True
    tus = np.array(recs, dtype=synth.HEVC_TU_DTYPE)
    res = np.concatenate(parts)
    got = ops.hevc_intra_recon(tus, res, w, h, False)[0]
    exp = O.oracle_hevc_intra(tus, res, w, h, False)[0]
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("tag", ["p1080", "p1080_constrained", "odd"])
def test_isp_picture_vs_reference(golden, tag):
    """tests/golden/hevc_isp.npz: 1080p-class pictures reconstructed by the reference's static intra_sample_prediction itself, with
    its own z-scan availability, picture edges that cut coding tree blocks included (test_oracle_golden.py::test_hevc_isp_picture)"""
    from test_oracle_golden import isp_digest, isp_inputs
    g = golden("hevc_isp.npz")
    w, h, seed, _, n = [int(x) for x in g[f"{tag}_spec"]]
    tus, res = isp_inputs(w, h, seed)
    got = ops.hevc_intra_recon(tus, res, w, h, True, 8, 8)
    assert np.array_equal(got[0][0], g[f"{tag}_row0"]) and np.array_equal(got[0][h - 1], g[f"{tag}_lastrow"])
    assert isp_digest(got) == g[f"{tag}_sha256"].tobytes()


@pytest.mark.parametrize("host_check", [False, True])
def test_one_bad_tu_in_a_large_list(host_check, monkeypatch):
    """Lists of 2^17 TUs and more: the host looks at a sample of the records (a bad one there is FFHIP_EINVAL at once), EVERY record is checked by
    a kernel in front of the planner, which refuses the call through the stream -- the call returns 0, nothing is written, ffhip_stream_sync says
    FFHIP_EINVAL once.  A TU that lies outside its plane, an availability bit that points outside, a bad size, in the last stretch of the list
    and in a stretch the host does not sample.  FFHIP_HEVC_HOST_CHECK=1: the whole list on host threads, as before round 4.  The untouched list
    decodes bit for bit afterwards."""
    if host_check:
        monkeypatch.setenv("FFHIP_HEVC_HOST_CHECK", "1"); capi.reload_env()
    L = capi.require_device()
    w, h = 3840, 2176
    tus, res = synth.hevc_intra_tus(w, h, seed=91)
    n = len(tus)
    assert n >= (1 << 17) + 1000
    exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8)
    dr = ops.DeviceBuffer(res)
    dy, du, dv = ops.DeviceBuffer(nbytes=w * h * 2), ops.DeviceBuffer(nbytes=w * h // 2), ops.DeviceBuffer(nbytes=w * h // 2)

    def call(t):
        dt = ops.DeviceBuffer(np.ascontiguousarray(t).view(np.uint8))     # the same records on both sides, as the entry point asks
        for d in (dy, du, dv):
            capi.check(L.ffhip_memset(d.ptr, 0x11, d.nbytes, None))
        rc = L.ffhip_hevc_intra_recon(t.ctypes.data, dt.ptr, len(t), dr.ptr, dy.ptr, du.ptr, dv.ptr, w, h, w, w // 2, h // 2, w // 2, 8, 8, None)
        rs = L.ffhip_stream_sync(None)
        return rc, rs
    unsampled = 5 * 4096 + 77                                             # stretch 5 of 4096 records: the host's sample (every 64th) skips it
    assert unsampled < n
    for field, value in (("x", w), ("log2_size", 6), ("pred_mode", 35), ("cidx", 3)):
        for at in (n - 7, unsampled):
            bad = tus.copy()
            bad[field][at] = value
            rc, rs = call(bad)
            assert (rc, rs) in ((capi.FFHIP_EINVAL, 0), (0, capi.FFHIP_EINVAL)), (field, at, rc, rs)
            if host_check:
                assert rc == capi.FFHIP_EINVAL
            assert L.ffhip_stream_sync(None) == 0
            assert (dy.to_host((w * h * 2,), np.uint8) == 0x11).all() and (du.to_host((w * h // 2,), np.uint8) == 0x11).all(), (field, at)
    bad = tus.copy()
    i = int(np.nonzero((bad["y"] == 0) & (bad["cidx"] == 0))[0][-1])      # a TU of the top row claims a row above
    bad["avail_top"][i] = 1
    rc, rs = call(bad)
    assert (rc, rs) in ((capi.FFHIP_EINVAL, 0), (0, capi.FFHIP_EINVAL))
    assert L.ffhip_stream_sync(None) == 0
    for d in (dy, du, dv):
        capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
    dt = ops.DeviceBuffer(tus.view(np.uint8))
    assert L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.ptr, n, dr.ptr, dy.ptr, du.ptr, dv.ptr, w, h, w, w // 2, h // 2, w // 2, 8, 8, None) == 0
    assert L.ffhip_stream_sync(None) == 0
    got = (dy.to_host((h, w), np.int16), du.to_host((h // 2, w // 2), np.int16), dv.to_host((h // 2, w // 2), np.int16))
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("w,h,seed,kw,keys", [(512, 512, 5, dict(tu_mix="c5"), True), (512, 384, 2, dict(), True), (1920, 1088, 2, dict(), True),
                                              (128, 128, 43, dict(ccp=True, chroma_444=True, adversarial_masks=True), None),
                                              (256, 192, 41, dict(adversarial_masks=True), None), (256, 128, 44, dict(ctb=32), False)])
def test_device_planner_takes_what_it_should(w, h, seed, kw, keys):
    """A list the device planner refuses is still decoded bit-exactly -- by ONE wave -- so parity alone would not notice a planner that
    refuses what it should take: the planner's verdict of the call (ffhip_debug_hevc_plan_result) says "taken", and for lists in coding-tree
    order of 64x64 blocks "tickets by wavefront" (a 32x32 coding tree block enters a 64x64 cell twice: decode order; lists with availability
    masks that point at samples decoded later may go either way)."""
    import ctypes as C
    tus, res = synth.hevc_intra_tus(w, h, seed, **kw)
    csub = 1 if kw.get("chroma_444") else 2
    got = ops.hevc_intra_recon(tus, res, w, h, True, 8, 8, csub=csub)
    out = (C.c_uint32 * 8)()
    capi.check(capi.lib().ffhip_debug_hevc_plan_result(out), "ffhip_debug_hevc_plan_result")
    assert out[6] == 0 and out[0] == 0, list(out)          # valid, and not left to the serial kernel
    assert out[1] > 0
    if keys is not None:
        assert (out[3] == 0) == keys, list(out)
    assert (out[4] > 0) == (out[3] == 0)                   # the widest wavefront is known exactly when there are wavefront tickets
    exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8, csub=csub)
    for gp, e, name in zip(got, exp, "YUV"):
        assert np.array_equal(gp, e), name


@pytest.mark.parametrize("w,h,seed,kw", [(512, 512, 5, dict(tu_mix="c5")), (512, 384, 2, dict()), (1920, 1088, 2, dict(tu_mix="c5")),
                                         (256, 192, 9, dict(min_tu=8))])
def test_reference_order_takes_the_wavefront_schedule(w, h, seed, kw):
    """The reference decodes coding unit by coding unit -- luma tree, Cb, Cr (coding/hevc.c:5013-5180) -- so its TU lists switch planes inside
    every 64x64 area.  The planner defines runs, window visits and cell visits on each plane's own subsequence (it works on the list sorted by
    plane): such a list gets the 64x64 windows, as many groups and the coding-tree wavefront tickets of the same list with the planes of a
    coding tree block one after the other -- and the same picture."""
    tus, res = synth.hevc_intra_tus(w, h, seed, **kw)
    ref = synth.hevc_reference_order(tus, 64, 2, seed)
    assert (ref["cidx"][1:] != ref["cidx"][:-1]).sum() > (tus["cidx"][1:] != tus["cidx"][:-1]).sum()
    exp = O.oracle_hevc_intra(ref, res, w, h, True, 8, 8)
    got_p = ops.hevc_intra_recon(tus, res, w, h, True, 8, 8)
    plan_p = assert_wavefront_schedule(sorted_by_plane=False)
    got_r = ops.hevc_intra_recon(ref, res, w, h, True, 8, 8)
    plan_r = assert_wavefront_schedule(sorted_by_plane=True)
    assert plan_r[1] == plan_p[1] and plan_r[4] == plan_p[4], (plan_p, plan_r)     # groups, widest wavefront
    for gp, gr, e, name in zip(got_p, got_r, exp, "YUV"):
        assert np.array_equal(gr, e), name
        assert np.array_equal(gp, e), name


def _raster_420_list(w, h, seed):
    """4x4 TUs of all three planes in RASTER order (no window has contiguous groups: the device planner refuses the list)"""
    rng = np.random.default_rng(seed)
    recs, parts, off = [], [], 0
    for c, (pw, ph) in enumerate(((w, h), (w // 2, h // 2), (w // 2, h // 2))):
        done = np.zeros((ph, pw), bool)
        for y0 in range(0, ph, 4):
            for x0 in range(0, pw, 4):
                at = al = 0
                for k in range(8):
                    if y0 > 0 and x0 + k < pw and done[y0 - 1, x0 + k]:
                        at |= 1 << k
                    if x0 > 0 and y0 + k < ph and done[y0 + k, x0 - 1]:
                        al |= 1 << k
                fl = synth.TU_RESIDUAL | (synth.TU_CORNER if x0 > 0 and y0 > 0 else 0)
                recs.append((x0, y0, 2, c, int(rng.integers(0, 35)), fl, off, 0, at, al))
                parts.append(np.rint(rng.laplace(0, 12, size=16)).astype(np.int16))
                off += 16
                done[y0:y0 + 4, x0:x0 + 4] = True
    return np.array(recs, dtype=synth.HEVC_TU_DTYPE), np.concatenate(parts)


@pytest.mark.parametrize("w,h,seed,kw,env", [(64, 64, 1, dict(), {}), (192, 128, 2, dict(adversarial_masks=True), {}), (512, 512, 5, dict(tu_mix="c5"), {}),
                                             (1920, 1080, 3, dict(), {}), (200, 104, 4, dict(ctb=8), {}),                                              (320, 192, 6, dict(), {"FFHIP_HEVC_PLAN": "host"}), (320, 192, 6, dict(), {"FFHIP_HEVC_INTRA_MODE": "levels"}),
                                             (320, 192, 6, dict(), {"FFHIP_HEVC_INTRA_WINDOW": "4"}), (320, 192, 6, dict(), {"FFHIP_HEVC_TILE_EARLY": "0"}),
                                             (256, 192, 7, dict(ctb=32), {}), (96, 48, 8, "raster", {"FFHIP_HEVC_PLAN": "device"})])
def test_decode_tiles_emits_the_colour_cell_by_cell(w, h, seed, kw, env, monkeypatch):
    """ffhip_hevc_decode_tiles: the planes of ffhip_hevc_intra_recon and the BGRA of YUV420_to_BGRA32_16bit over them (the oracle's restatement), for
    pictures of any 4:2:0 size, whichever schedule the list takes (the device planner's, the host's, levels, the serial kernel of a refused list)"""
    from test_color_gpu import oracle_420_16
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    capi.reload_env()
    if kw == "raster":
        tus, res = _raster_420_list(w, h, seed)
    else:
        tus, res = synth.hevc_intra_tus(w, h, seed, **kw)
    exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8)
    want = oracle_420_16(exp[0], exp[1], exp[2], h // 2, w // 2, 2)
    for rep in range(2):
        bgra, planes = ops.hevc_decode_tiles(tus, res, w, h)
        for gp, e, name in zip(planes, exp, "YUV"):
            assert np.array_equal(gp, e), (name, rep)
        assert np.array_equal(bgra, want), (rep, np.argwhere(bgra != want)[:4])
    for k in env:
        monkeypatch.delenv(k)
    capi.reload_env()


def test_decode_tiles_padded_pitch_and_reference_order():
    tus, res = synth.hevc_intra_tus(384, 256, 12, tu_mix="c5")
    ref = synth.hevc_reference_order(tus, 64, 2, 12)
    a, _ = ops.hevc_decode_tiles(tus, res, 384, 256)
    b, _ = ops.hevc_decode_tiles(ref, res, 384, 256, pitch=384 * 4 + 1024)
    assert np.array_equal(b[:, :384 * 4], a) and not b[:, 384 * 4:].any()


def test_tile_call_refuses_a_bad_record_and_recovers():
    """ffhip_hevc_intra_recon_tiles with one bad record in a large list: the validation kernel runs on the library's plan stream, the refusal travels through the
    planner's result words to the grouped kernel on the caller's stream, ffhip_stream_sync says FFHIP_EINVAL once and nothing is written; the calls behind it (the
    other scratch set, then the first again) decode the untouched list bit for bit"""
    L = capi.require_device()
    w, h = 3840, 2176
    tus, res = synth.hevc_intra_tus(w, h, seed=91)
    n = len(tus)
    assert n >= (1 << 17) + 1000
    exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8)
    dr = ops.DeviceBuffer(res)
    dy, du, dv = ops.DeviceBuffer(nbytes=w * h * 2), ops.DeviceBuffer(nbytes=w * h // 2), ops.DeviceBuffer(nbytes=w * h // 2)
    tf = np.zeros(1, np.int64)

    def call(t):
        dt = ops.DeviceBuffer(np.ascontiguousarray(t).view(np.uint8))
        for d in (dy, du, dv):
            capi.check(L.ffhip_memset(d.ptr, 0x11, d.nbytes, None))
        capi.check(L.ffhip_stream_sync(None))
        rc = L.ffhip_hevc_intra_recon_tiles(t.ctypes.data, dt.ptr, len(t), tf.ctypes.data, 1, dr.ptr, dy.ptr, du.ptr, dv.ptr, w, h, w, w // 2, h // 2, w // 2, 8, 8, None)
        return rc, L.ffhip_stream_sync(None), dt
    bad = tus.copy()
    bad["x"][5 * 4096 + 77] = w                       # a stretch the host's sample skips: the device finds it
    rc, rs, _ = call(bad)
    assert (rc, rs) in ((capi.FFHIP_EINVAL, 0), (0, capi.FFHIP_EINVAL)), (rc, rs)
    assert L.ffhip_stream_sync(None) == 0
    assert (dy.to_host((w * h * 2,), np.uint8) == 0x11).all() and (du.to_host((w * h // 2,), np.uint8) == 0x11).all()
    for rep in range(3):
        rc, rs, _ = call(tus)
        assert (rc, rs) == (0, 0), rep
        got = (dy.to_host((h, w), np.int16), du.to_host((h // 2, w // 2), np.int16), dv.to_host((h // 2, w // 2), np.int16))
        for a, b in zip(got, exp):
            assert np.array_equal(a, b), rep


def test_tile_calls_alternating_between_two_streams():
    """The one-chunk tile call keeps two scratch sets per CALLER'S stream and guards each with an event of that stream (round 6; the guard was the calling
    thread's before).  One thread, streams A, B, B, A, A, five different lists into five plane sets, no sync in between: the fifth call's pre-pass runs on
    the library's plan stream and must wait for the first call's grouped kernel on A -- it takes that call's scratch.  Every plane set equals the oracle."""
    L = capi.require_device()
    w, h = 1024, 512
    sa, sb = L.ffhip_stream_create(), L.ffhip_stream_create()
    assert sa and sb
    order = [sa, sb, sb, sa, sa]
    tf = np.zeros(1, np.int64)
    keep, jobs = [], []
    for k, st in enumerate(order):
        tus, res = synth.hevc_intra_tus(w, h, seed=300 + k, tu_mix="c5")
        t = np.ascontiguousarray(tus)
        dt, dr = ops.DeviceBuffer(t.view(np.uint8)), ops.DeviceBuffer(res)
        pl = (ops.DeviceBuffer(nbytes=w * h * 2), ops.DeviceBuffer(nbytes=w * h // 2), ops.DeviceBuffer(nbytes=w * h // 2))
        for d in pl:
            capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
        keep.append((t, dt, dr))
        jobs.append((tus, res, pl))
    capi.check(L.ffhip_stream_sync(None))
    for (t, dt, dr), (_, _, pl), st in zip(keep, jobs, order):
        capi.check(L.ffhip_hevc_intra_recon_tiles(t.ctypes.data, dt.ptr, len(t), tf.ctypes.data, 1, dr.ptr, pl[0].ptr, pl[1].ptr, pl[2].ptr, w, h, w, w // 2, h // 2, w // 2,
                                                  8, 8, st))
    assert L.ffhip_stream_sync(sa) == 0 and L.ffhip_stream_sync(sb) == 0
    for k, (tus, res, pl) in enumerate(jobs):
        exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8)
        got = (pl[0].to_host((h, w), np.int16), pl[1].to_host((h // 2, w // 2), np.int16), pl[2].to_host((h // 2, w // 2), np.int16))
        for a, b in zip(got, exp):
            assert np.array_equal(a, b), k
    L.ffhip_release_caches()                       # the guards go with the scratches; the next call makes new ones
    t, dt, dr = keep[0]
    pl = jobs[0][2]
    capi.check(L.ffhip_hevc_intra_recon_tiles(t.ctypes.data, dt.ptr, len(t), tf.ctypes.data, 1, dr.ptr, pl[0].ptr, pl[1].ptr, pl[2].ptr, w, h, w, w // 2, h // 2, w // 2, 8, 8, sa))
    assert L.ffhip_stream_sync(sa) == 0
    L.ffhip_stream_destroy(sa)
    L.ffhip_stream_destroy(sb)
