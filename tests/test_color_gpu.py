"""GPU: planar YUV->BGRA converters against goldens and the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import ops

pytestmark = pytest.mark.gpu


def oracle_420_8(y, u, v, mbr, mbc):
    H, W = y.shape
    o = np.zeros((H, W * 4), np.uint8)
    O.ffo().ffo_yuv420_to_bgra32(o.reshape(-1), W * 4, np.ascontiguousarray(y).reshape(-1),
                                 np.ascontiguousarray(u).reshape(-1), np.ascontiguousarray(v).reshape(-1), W, W // 2, mbr, mbc)
    return o


def oracle_420_16(y, u, v, r, c, ctb):
    H, W = y.shape
    o = np.zeros((H, W * 4), np.uint8)
    O.ffo().ffo_yuv420_to_bgra32_16bit(o.reshape(-1), W * 4, np.ascontiguousarray(y).reshape(-1),
                                       np.ascontiguousarray(u).reshape(-1), np.ascontiguousarray(v).reshape(-1),
                                       W, W // 2, r, c, ctb)
    return o


def test_golden_planar(golden):
    g = golden("color_planar.npz")
    assert np.array_equal(ops.yuv420_to_bgra(g["p420_y"][None], g["p420_u"][None], g["p420_v"][None], 3, 4)[0], g["p420_bgra"])
    assert np.array_equal(ops.yuv420_to_bgra_16(g["p16_y"][None], g["p16_u"][None], g["p16_v"][None], 3, 4, 16)[0], g["p16_bgra"])
    assert np.array_equal(ops.yuv400_to_bgra_16(g["p16_y"][None], 3, 4, 16)[0], g["p400_bgra"])


def test_webp_planes_all_chroma_pairs():
    """every (u,v) in [0,255]^2 appears (so every exact-integer-G pair of the 8-bit domain), 3 images"""
    rng = np.random.default_rng(0)
    mbr, mbc, n = 32, 32, 3                       # 512x512: chroma 256x256 = all pairs
    uu, vv = np.meshgrid(np.arange(256), np.arange(256))
    u = np.stack([uu.astype(np.uint8)] * n)
    v = np.stack([vv.astype(np.uint8)] * n)
    y = rng.integers(0, 256, size=(n, 512, 512)).astype(np.uint8)
    got = ops.yuv420_to_bgra(y, u, v, mbr, mbc)
    for i in range(n):
        assert np.array_equal(got[i], oracle_420_8(y[i], u[i], v[i], mbr, mbc)), i


def test_hevc_planes_domains_and_ctb_sizes():
    rng = np.random.default_rng(1)
    for ctb, r, c in ((16, 3, 5), (32, 2, 3), (64, 1, 2)):
        H, W = ctb * r, ctb * c
        for lo, hi in ((0, 256), (0, 1024), (0, 8192), (-32768, 32768)):   # 8/10-bit, IDCT range, wrapped garbage
            y = rng.integers(lo, hi, size=(2, H, W)).astype(np.int16)
            u = rng.integers(lo, hi, size=(2, H // 2, W // 2)).astype(np.int16)
            v = rng.integers(lo, hi, size=(2, H // 2, W // 2)).astype(np.int16)
            got = ops.yuv420_to_bgra_16(y, u, v, r, c, ctb)
            g400 = ops.yuv400_to_bgra_16(y, r, c, ctb)
            for i in range(2):
                assert np.array_equal(got[i], oracle_420_16(y[i], u[i], v[i], r, c, ctb)), (ctb, lo, hi, i)
                o = np.zeros((H, W * 4), np.uint8)
                O.ffo().ffo_yuv400_to_bgra32_16bit(o.reshape(-1), W * 4, np.ascontiguousarray(y[i]).reshape(-1), W, r, c, ctb)
                assert np.array_equal(g400[i], o)


def test_exact_integer_green_pairs_16bit(golden):
    tri = golden("color_triples.npz")["yuv"]
    uu, vv = tri[:, 1].astype(np.int64) - 128, tri[:, 2].astype(np.int64) - 128
    sel = ((215 * uu + 381 * vv) % 1000 == 0) & (tri[:, 1] >= 0) & (tri[:, 2] >= 0) & (tri[:, 0] >= 0)
    t = tri[sel][:64 * 64]
    n = len(t) // 64 * 64
    t = t[:n]
    # one chroma sample per 2x2 luma; place pair k at chroma (k // 64, k % 64) of a 128 x 128 picture
    rows = n // 64
    u = t[:, 1].reshape(rows, 64)
    v = t[:, 2].reshape(rows, 64)
    y = np.repeat(np.repeat(t[:, 0].reshape(rows, 64), 2, axis=0), 2, axis=1)
    pad = (-rows) % 8
    if pad:
        u = np.pad(u, ((0, pad), (0, 0)), constant_values=128); v = np.pad(v, ((0, pad), (0, 0)), constant_values=128)
        y = np.pad(y, ((0, 2 * pad), (0, 0)))
    r = y.shape[0] // 16
    got = ops.yuv420_to_bgra_16(y[None].astype(np.int16), u[None].astype(np.int16), v[None].astype(np.int16), r, 8, 16)[0]
    assert np.array_equal(got, oracle_420_16(y.astype(np.int16), u.astype(np.int16), v.astype(np.int16), r, 8, 16))
