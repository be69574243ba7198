"""GPU: the per-block, reference-shaped entry points (accl_ops / dct_ops / cs_ops /
idct_4x4_hevc) are bit-exact with the golden vectors, and the registration protocol
of arch/accl.c works against the real reference registry when oracle/_ref travelled."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops

pytestmark = pytest.mark.gpu


def test_idct_8x8_table(golden):
    g = golden("jpeg_blocks.npz")
    idx = list(range(0, 40)) + list(range(257, 513, 8))
    for i in idx:
        b = g["coef"][i].copy()
        ops.idct_8x8(b)
        assert np.array_equal(b, g["idct"][i]), i


def test_idct_4x4_vp8_table(golden):
    g = golden("vp8_blocks.npz")
    for i in list(range(0, 32)) + list(range(256, 512, 8)):
        b = g["coef"][i].copy()
        ops.idct_4x4(b)
        assert np.array_equal(b, g["idct"][i]), i


def test_idct_4x4_hevc(golden):
    g = golden("hevc_dst4.npz")
    for bd in (8, 10):
        for epp in (0, 1):
            for i in list(range(0, 16)) + list(range(128, 256, 8)):
                out = ops.idct_4x4_hevc(g["coef"][i].copy(), bd, bool(epp))
                assert np.array_equal(out, g[f"dst_bd{bd}_epp{epp}"][i]), (bd, epp, i)


def test_cs_ops_mcu(golden):
    g = golden("color_planar.npz")
    for (v, h) in ((1, 1), (1, 2), (2, 1), (2, 2), (1, 4), (4, 1), (1, 3), (3, 1)):
        out = ops.yuv_to_bgra32(g["mcu_Y"], g["mcu_U"], g["mcu_V"], v, h)
        assert np.array_equal(out, g[f"mcu_v{v}h{h}"]), (v, h)
    tri = golden("color_triples.npz")
    for i in range(0, 64 * 40, 64):
        out = ops.yuv_to_bgra32(np.ascontiguousarray(tri["yuv"][i:i + 64, 0]), np.ascontiguousarray(tri["yuv"][i:i + 64, 1]),
                                np.ascontiguousarray(tri["yuv"][i:i + 64, 2]), 1, 1)
        assert np.array_equal(out.reshape(64, 4), tri["bgra"][i:i + 64])


def test_accl_ops_struct_and_registration():
    L = capi.require_device()
    p = L.ffhip_accl_ops_get()
    assert p and p.contents.type == 27
    blk = np.arange(16, dtype=np.int16)
    exp = blk.copy()
    O.ffo().ffo_vp8_idct_4x4(exp)
    p.contents.idct_4x4(blk.ctypes.data, 8)
    assert np.array_equal(blk, exp)
    if not os.path.exists(O.REF_SO):
        pytest.skip("oracle/_ref did not travel: registration against the real registry not checked")
    R = C.CDLL(O.REF_SO, mode=C.RTLD_GLOBAL)      # the reference's own arch/accl.c registry
    R.accl_find.restype = C.c_void_p
    assert not R.accl_find(27)
    L.hip_accl_init()                             # finds accl_ops_register in the process and registers
    found = R.accl_find(27)
    assert found == C.addressof(p.contents)
    # the reference's consumer-side call shape: ops->idct_8x8(block, bitdepth)
    ops8 = C.cast(found, C.POINTER(capi.AcclOps)).contents
    b8 = np.zeros(64, dtype=np.int16); b8[0] = 873; b8[1] = 55
    e8 = b8.copy()
    O.ffo().ffo_idct_8x8_16(e8)
    ops8.idct_8x8(b8.ctypes.data, 8)
    assert np.array_equal(b8, e8)
    L.hip_accl_uninit()
