"""Operator-level host mirror of the reference interface for the hot path.

Names and argument meaning follow the reference (format/jpg.c:540-560,
utils/idct.h:14-25, utils/colorspace.h:29-33, arch/accl.h:20-25); every function
runs on the GPU through the C ABI of libffpic_hip.so and raises FfhipError when
the library or a gfx950 device is missing (no CPU fallback, by design).
"""
import ctypes as C

import numpy as np

from . import capi


def _vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def jpeg_recon_batch_host(geom, n_images, coef_y, coef_u, coef_v, quant):
    """dequant + idct_8x8 + YUV_to_BGRA32 for a batch held in host numpy arrays
    (the per-picture call a patched format/jpg.c would make).  Returns BGRA
    uint8 [n, H, W, 4]."""
    L = capi.require_device()
    H, W = geom.height, geom.width
    out = np.empty((n_images, H, W, 4), dtype=np.uint8)
    quant = np.ascontiguousarray(quant, dtype=np.uint16)
    qstride = 0 if quant.ndim == 2 else 256
    capi.check(L.ffhip_jpeg_recon_batch_host(C.byref(geom), n_images, _vp(coef_y), _vp(coef_u), _vp(coef_v),
                                              _vp(quant), qstride, _vp(out), W * 4, H * W * 4),
               "ffhip_jpeg_recon_batch_host")
    return out


def jpeg_recon_batch(geom, n_images, d_coef_y, d_coef_u, d_coef_v, d_quant, quant_stride, d_bgra, pitch,
                     image_stride, d_workspace=None, workspace_bytes=0, stream=None):
    """Device-pointer form (ints / c_void_p); only enqueues on `stream`."""
    L = capi.lib()
    capi.check(L.ffhip_jpeg_recon_batch(C.byref(geom), n_images, d_coef_y, d_coef_u, d_coef_v, d_quant,
                                        quant_stride, d_bgra, pitch, image_stride, d_workspace, workspace_bytes,
                                        stream), "ffhip_jpeg_recon_batch")


def idct_8x8(block, bitdepth=8):
    """get_dct_ops(16)->idct_8x8 (utils/idct.c:512-534): in-place on int16[64]."""
    L = capi.require_device()
    ops = L.ffhip_get_dct_ops(16)
    if not ops:
        raise capi.FfhipError("ffhip_get_dct_ops(16) returned NULL")
    assert block.dtype == np.int16 and block.size == 64 and block.flags.c_contiguous
    ops.contents.idct_8x8(block.ctypes.data, bitdepth)
    return block


def idct_4x4(block, bitdepth=8):
    """get_dct_ops(16)->idct_4x4 == VP8 IDCT (utils/idct.c:100-151), in place."""
    L = capi.require_device()
    ops = L.ffhip_get_dct_ops(16)
    if not ops:
        raise capi.FfhipError("ffhip_get_dct_ops(16) returned NULL")
    assert block.dtype == np.int16 and block.size == 16 and block.flags.c_contiguous
    ops.contents.idct_4x4(block.ctypes.data, bitdepth)
    return block


def idct_4x4_hevc(block, bitdepth=8, epp=False):
    """idct_4x4_hevc (utils/idct.c:36-55)."""
    L = capi.require_device()
    out = np.empty(16, dtype=np.int16)
    L.ffhip_idct_4x4_hevc(block.ctypes.data, out.ctypes.data, bitdepth, epp)
    return out


def yuv_to_bgra32(Y, U, V, v, h, pitch=None):
    """get_cs_ops(16)->YUV_to_BGRA32 for one MCU (utils/colorspace.c:133-172)."""
    L = capi.require_device()
    ops = L.ffhip_get_cs_ops(16)
    if not ops:
        raise capi.FfhipError("ffhip_get_cs_ops(16) returned NULL")
    pitch = pitch or 8 * h * 4
    out = np.zeros((8 * v, pitch), dtype=np.uint8)
    ops.contents.YUV_to_BGRA32(out.ctypes.data, pitch, Y.ctypes.data, U.ctypes.data, V.ctypes.data, v, h)
    return out
