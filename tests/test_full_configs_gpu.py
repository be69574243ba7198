"""GPU: BASELINE.json's configurations at their FULL sizes, bit-for-bit.

  C2  1024 x 1920x1088 4:2:0 coefficient grids on one MI355X        (configs[1])
  C3  3840x2160 grids; 264 images, so that BOTH the BGRA output (33.2 MB per image: byte offset 2^32 falls inside
      image 129) and the luma coefficient plane (16.6 MB per image: 2^32 falls inside image 258) cross 4 GiB in one
      launch                                                         (configs[2]; 256 images do not cross in the plane)
  C5  one 7680x4352 HEVC intra picture (8K coded to whole 64x64 coding tree blocks), both the random quadtree down to
      4x4 and the TU mix SURVEY 8d names for config 5, plus a 48-tile grid                     (configs[4])

The batches are built on the device from K unique images in a seeded, non-periodic order (so a kernel that fetched
image i from the slot of another image could not pass), every unique image's output is compared with the C oracle,
and every image of the batch with the oracle-checked output of its unique source -- all bytes, not a checksum.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth

pytestmark = pytest.mark.gpu


def _jpeg_full_batch(cols, rows, n, n_unique, seed, must_cross=(), h=2, v=2):
    torch = pytest.importorskip("torch")
    capi.require_device()
    dev = torch.device("cuda:0")
    geom = O.make_geom(cols, rows, 3, h, v)
    cg = capi.jpeg_geom(cols, rows, 3, h, v)
    H, W = geom.height, geom.width
    mcus = cols * rows
    q = synth.quant_tables()
    cy, cu, cv = synth.coef_batch(n_unique, cols, rows, h=h, v=v, first=seed)
    exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n_unique, n_threads=4)
    rng = np.random.default_rng(seed)
    order = rng.integers(0, n_unique, size=n)
    order[:n_unique] = np.arange(n_unique)                 # every unique image appears
    order[-1] = n_unique - 1
    idx = torch.from_numpy(order).to(dev)
    t_y = torch.from_numpy(cy).to(dev).view(n_unique, -1)[idx].reshape(-1)
    t_u = torch.from_numpy(cu).to(dev).view(n_unique, -1)[idx].reshape(-1)
    t_v = torch.from_numpy(cv).to(dev).view(n_unique, -1)[idx].reshape(-1)
    t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
    pitch, stride = W * 4, W * 4 * H
    for what, per_image in must_cross:
        assert n * per_image > 1 << 32, what
    out = torch.full((n * stride,), 0x5A, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ops.jpeg_recon_batch(cg, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0, out.data_ptr(),
                         pitch, stride, None, 0, st)
    torch.cuda.synchronize()
    o = out.view(n, stride)
    # the unique images, wherever they first sit, against the oracle
    for k in range(n_unique):
        assert np.array_equal(o[k].cpu().numpy().reshape(H, W, 4), exp[k]), ("unique image", k)
    e_dev = torch.from_numpy(exp.reshape(n_unique, -1)).to(dev)
    bad = [i for i in range(n) if not torch.equal(o[i], e_dev[int(order[i])])]
    assert not bad, f"{len(bad)} of {n} images differ from the oracle's picture, first {bad[:8]}"
    return o, exp, order, (H, W, mcus)


def test_c2_1024_images_1080p():
    """BASELINE configs[1]: the whole batch of 1024 (15 GB in + out), every image against the oracle's bytes"""
    _jpeg_full_batch(120, 68, 1024, 4, seed=2000)


def test_c3_4k_batch_crossing_4gib():
    """BASELINE configs[2] geometry (240 x 135 MCUs) with 264 images: first, last, the image straddling byte 2^32 of the
    output (129) and the one straddling byte 2^32 of the luma coefficient plane (258) are ordinary members of the
    all-images comparison; they are re-checked against the oracle on the host for the record."""
    cols, rows, n = 240, 135, 264
    o, exp, order, (H, W, mcus) = _jpeg_full_batch(cols, rows, n, 4, seed=3000,
                                                   must_cross=(("BGRA", 3840 * 2160 * 4), ("luma plane", 240 * 135 * 4 * 128)))
    out_straddle = (1 << 32) // (W * 4 * H)
    coef_straddle = (1 << 32) // (mcus * 4 * 128)
    assert (out_straddle, coef_straddle) == (129, 258)
    for i in (0, out_straddle, coef_straddle, n - 1):
        assert np.array_equal(o[i].cpu().numpy().reshape(H, W, 4), exp[int(order[i])]), i


@pytest.mark.parametrize("h,v", [(4, 1), (1, 4), (1, 1)])
def test_4k_batches_of_the_other_layouts(h, v):
    """The MCU layouts round 3 added (4:1:1 = h4v1 and its transpose) and 4:4:4 at the headline's picture size, 64 images in one
    launch of the fused strip kernel: every image against the oracle's bytes (3840 is 120 MCUs of 32 pixels; 2160 is not a whole
    number of 32-line MCUs, so the h1v4 pictures are 2176 lines)"""
    cols, rows = 3840 // (8 * h), -(-2160 // (8 * v))
    _jpeg_full_batch(cols, rows, 64, 3, seed=4100 + 10 * h + v, h=h, v=v)


def _intra_full_picture(W, H, tus, res, envs=({},), monkeypatch=None, sorted_by_plane=None, tile_first=None, exp=None, bgra_exp=None):
    """... once per entry of `envs` (library switches for the call: the oracle's picture is worked out once).  tile_first: through
    ffhip_hevc_intra_recon_tiles (the list is the concatenation of independent tiles starting at these records)"""
    torch = pytest.importorskip("torch")
    L = capi.require_device()
    dev = torch.device("cuda:0")
    if exp is None:
        exp = O.oracle_hevc_intra(tus, res, W, H, True, 8, 8)
    tf = None if tile_first is None else np.ascontiguousarray(tile_first, dtype=np.int64)
    dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev)
    dr = torch.from_numpy(res).to(dev)
    py = torch.zeros((H, W), dtype=torch.int16, device=dev)
    pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev)
    pv = torch.zeros_like(pu)
    st = torch.cuda.current_stream().cuda_stream
    for env in envs:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        capi.reload_env()
        for rep in range(2):                               # the second call reuses the per-stream scratch and schedule buffers
            for p in (py, pu, pv):
                p.zero_()
            if bgra_exp is not None:
                out = torch.zeros((H, W * 4), dtype=torch.uint8, device=dev)
                capi.check(L.ffhip_hevc_decode_tiles(tus.ctypes.data, dt.data_ptr(), len(tus), tf.ctypes.data, len(tf), dr.data_ptr(), py.data_ptr(), pu.data_ptr(),
                                                     pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, out.data_ptr(), W * 4, st), "ffhip_hevc_decode_tiles")
            elif tf is None:
                capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), len(tus), dr.data_ptr(), py.data_ptr(), pu.data_ptr(),
                                                    pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st), "ffhip_hevc_intra_recon")
            else:
                capi.check(L.ffhip_hevc_intra_recon_tiles(tus.ctypes.data, dt.data_ptr(), len(tus), tf.ctypes.data, len(tf), dr.data_ptr(), py.data_ptr(), pu.data_ptr(),
                                                          pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st), "ffhip_hevc_intra_recon_tiles")
            capi.check(L.ffhip_stream_sync(st), "sync")
            verdict = (C.c_uint32 * 8)()   # the device planner took the list, with wavefront tickets: not the one-wave serial path
            capi.check(L.ffhip_debug_hevc_plan_result(verdict), "ffhip_debug_hevc_plan_result")
            assert verdict[0] == 0 and verdict[3] == 0 and verdict[6] == 0, (env, list(verdict))
            assert verdict[5] == 6 and (sorted_by_plane is None or bool(verdict[7]) == sorted_by_plane), (env, list(verdict))
            for got, e, name in zip((py, pu, pv), exp, "YUV"):
                g = got.cpu().numpy()
                if not np.array_equal(g, e):
                    ys, xs = np.nonzero(g != e)
                    raise AssertionError(f"{name} plane differs in {len(ys)} samples (pass {rep}, {env}), first at x={xs[0]} y={ys[0]}")
            if bgra_exp is not None:
                g = out.cpu().numpy()
                if not np.array_equal(g, bgra_exp):
                    ys, xs = np.nonzero(g != bgra_exp)
                    raise AssertionError(f"BGRA differs in {len(ys)} bytes (pass {rep}, {env}), first at x={xs[0] // 4} y={ys[0]}")
        for k in env:
            monkeypatch.delenv(k)
        capi.reload_env()
    return exp


@pytest.mark.parametrize("mix,seed", [(None, 2), ("c5", 5)])
def test_c5_8k_intra_picture(mix, seed):
    """BASELINE configs[4]: ffhip_hevc_intra_recon on one 7680x4352 picture against ffo_hevc_intra_recon over the whole
    planes.  mix=None is the random quadtree down to 4x4 (~700 k TUs: the device planner's radix sort and scans and the
    32-bit offset guard at full size); "c5" is SURVEY 8d's mix (luma 32/16 at 60/40, chroma 16/8)."""
    W, H = 7680, 4352
    tus, res = synth.hevc_intra_tus(W, H, seed=seed, tu_mix=mix)
    assert len(tus) > (600_000 if mix is None else 40_000)
    exp = _intra_full_picture(W, H, tus, res, sorted_by_plane=False)
    # the same picture from the list in the reference's order: per coding unit the luma tree, then Cb, then Cr (coding/hevc.c:5013-5180)
    exp_r = _intra_full_picture(W, H, synth.hevc_reference_order(tus, 64, 2, seed), res, sorted_by_plane=True)
    for a, b in zip(exp, exp_r):
        assert np.array_equal(a, b)


def test_c5_48_tile_grid():
    """48 independent 512x512 pictures (a 12-megapixel HEIF grid) side by side in one plane set, one call"""
    T, K = 512, 48
    tus0, res0 = synth.hevc_intra_tus(T, T, seed=3)
    tus = np.concatenate([tus0.copy() for _ in range(K)])
    for i in range(K):
        sl = slice(i * len(tus0), (i + 1) * len(tus0))
        tus["x"][sl] += np.where(tus0["cidx"] == 0, T * i, T // 2 * i).astype(np.uint16)
        tus["res_offset"][sl] += len(res0) * i
        # the tiles are independent pictures: nothing to the left of a tile's first column is available
    res = np.tile(res0, K)
    exp = _intra_full_picture(T * K, T, tus, res)
    for i in range(1, K):                                  # and every tile is the same picture
        assert np.array_equal(exp[0][:, i * T:(i + 1) * T], exp[0][:, :T])


def test_c5_135_tile_8k_grid(monkeypatch):
    """BASELINE configs[4] reads "single 8K tile grid": an 8K picture as a HEIF grid of 15 x 9 = 135 independent 512x512 tiles
    (7680x4608; the tile loop this replaces is heif.c:297-309), all tiles in ONE plane set and ONE ffhip_hevc_intra_recon call, against
    the oracle over the whole plane set; every tile is the same picture.  As shipped, with the throughput instance of the grouped kernel
    forced (larger grids take it by themselves), with one and eight ticket counters, and with each piece of the two-stream pre-pass in line."""
    T, gx, gy = 512, 15, 9
    t0, res0 = synth.hevc_intra_tus(T, T, seed=6)
    K = gx * gy
    tus = np.tile(t0, K)
    k = np.repeat(np.arange(K), len(t0))
    sc = np.where(tus["cidx"] == 0, T, T // 2)
    tus["x"] = (tus["x"].astype(np.int64) + (k % gx) * sc).astype(np.uint16)
    tus["y"] = (tus["y"].astype(np.int64) + (k // gx) * sc).astype(np.uint16)
    tus["res_offset"] += (k * len(res0)).astype(np.uint32)
    def grid_of(t0):
        tus = np.tile(t0, K)
        k = np.repeat(np.arange(K), len(t0))
        sc = np.where(tus["cidx"] == 0, T, T // 2)
        tus["x"] = (tus["x"].astype(np.int64) + (k % gx) * sc).astype(np.uint16)
        tus["y"] = (tus["y"].astype(np.int64) + (k // gx) * sc).astype(np.uint16)
        tus["res_offset"] += (k * len(res0)).astype(np.uint32)
        return tus
    exp = _intra_full_picture(T * gx, T * gy, tus, np.tile(res0, K), envs=({}, {"FFHIP_HEVC_INTRA_TP_WIDTH": "1"}, {"FFHIP_HEVC_TICKET_SHARDS": "1"}, {"FFHIP_HEVC_TICKET_SHARDS": "8"},
                                    {"FFHIP_HEVC_SWEEP_INLINE": "1"}, {"FFHIP_HEVC_DEPTH_DIAGONALS": "1"}, {"FFHIP_HEVC_PROGRAMS_INLINE": "1"},
                                    {"FFHIP_HEVC_JT_INLINE": "1"}),
                              monkeypatch=monkeypatch, sorted_by_plane=False)
    assert np.array_equal(grid_of(t0), tus)
    # every tile's list in the reference's order (per coding unit: luma, Cb, Cr): sorted by plane on the device, same picture
    exp_r = _intra_full_picture(T * gx, T * gy, grid_of(synth.hevc_reference_order(t0, 64, 2, 6)), np.tile(res0, K),
                                envs=({}, {"FFHIP_HEVC_JT_INLINE": "1"}), monkeypatch=monkeypatch, sorted_by_plane=True)
    for a, b in zip(exp, exp_r):
        assert np.array_equal(a, b)
    # the tile loop as a pipeline (ffhip_hevc_intra_recon_tiles): the list cut at tile boundaries into 1 .. 4 chunks, each chunk's pre-pass next to
    # the chunk before's grouped kernel -- in both orders of the records
    tile_first = np.arange(K, dtype=np.int64) * len(t0)
    _intra_full_picture(T * gx, T * gy, tus, np.tile(res0, K), envs=({}, {"FFHIP_HEVC_TILE_EARLY": "0"}, {"FFHIP_HEVC_TILE_SCRATCHES": "1"}, {"FFHIP_HEVC_TILE_CHUNKS": "2", "FFHIP_HEVC_TILE_WAVES_PCT": "50"}, {"FFHIP_HEVC_TILE_CHUNKS": "3"}, {"FFHIP_HEVC_TILE_CHUNKS": "4"},
                                                                       {"FFHIP_HEVC_TILE_CHUNKS": "4", "FFHIP_HEVC_JT_INLINE": "1"}),
                        monkeypatch=monkeypatch, tile_first=tile_first, exp=exp)
    _intra_full_picture(T * gx, T * gy, grid_of(synth.hevc_reference_order(t0, 64, 2, 6)), np.tile(res0, K), envs=({"FFHIP_HEVC_TILE_CHUNKS": "4"},),
                        monkeypatch=monkeypatch, tile_first=tile_first, exp=exp, sorted_by_plane=True)
    # ... and with the colour conversion in the same call (ffhip_hevc_decode_tiles)
    from test_color_gpu import oracle_420_16
    bgra_exp = oracle_420_16(exp[0], exp[1], exp[2], T * gy // 2, T * gx // 2, 2)
    _intra_full_picture(T * gx, T * gy, tus, np.tile(res0, K), envs=({}, {"FFHIP_HEVC_TILE_CHUNKS": "3"}, {"FFHIP_HEVC_TILE_EARLY": "0"}),
                        monkeypatch=monkeypatch, tile_first=tile_first, exp=exp, bgra_exp=bgra_exp)
    for i in range(1, K):
        ox, oy = (i % gx) * T, (i // gx) * T
        assert np.array_equal(exp[0][oy:oy + T, ox:ox + T], exp[0][:T, :T]), i
