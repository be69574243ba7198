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


class DeviceBuffer:
    """Tiny RAII wrapper over ffhip_malloc/ffhip_free for host-driven calls (no torch)."""

    def __init__(self, host=None, nbytes=None):
        L = capi.require_device()
        self.nbytes = host.nbytes if host is not None else nbytes
        self.ptr = L.ffhip_malloc(max(self.nbytes, 16))
        if not self.ptr:
            raise capi.FfhipError("ffhip_malloc failed")
        if host is not None and host.nbytes:
            capi.check(L.ffhip_memcpy_h2d(self.ptr, host.ctypes.data, host.nbytes, None), "h2d")
            capi.check(L.ffhip_stream_sync(None))

    def to_host(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        L = capi.lib()
        capi.check(L.ffhip_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None), "d2h")
        capi.check(L.ffhip_stream_sync(None))
        return out

    def __del__(self):
        try:
            capi.lib().ffhip_free(self.ptr)
        except Exception:
            pass


def yuv420_to_bgra(y, u, v, mbrows, mbcols, pitch=None):
    """YUV420_to_BGRA32 (utils/colorspace.c:291-329): uint8 planes [n][H][W], [n][H/2][W/2]."""
    L = capi.require_device()
    n, H, W = y.shape
    assert (H, W) == (16 * mbrows, 16 * mbcols) and u.shape == (n, H // 2, W // 2) == v.shape
    pitch = pitch or W * 4
    dy, du, dv = DeviceBuffer(np.ascontiguousarray(y)), DeviceBuffer(np.ascontiguousarray(u)), DeviceBuffer(np.ascontiguousarray(v))
    do = DeviceBuffer(nbytes=n * H * pitch)
    capi.check(L.ffhip_memset(do.ptr, 0, do.nbytes, None))
    capi.check(L.ffhip_yuv420_to_bgra(do.ptr, pitch, dy.ptr, du.ptr, dv.ptr, W, W // 2, mbrows, mbcols, n,
                                      H * W, H * W // 4, H * pitch, None), "ffhip_yuv420_to_bgra")
    return do.to_host((n, H, pitch), np.uint8)


def yuv420_to_bgra_16(y, u, v, ctbrows, ctbcols, ctbsize, pitch=None):
    """YUV420_to_BGRA32_16bit (utils/colorspace.c:628-669): int16 planes."""
    L = capi.require_device()
    n, H, W = y.shape
    assert (H, W) == (ctbsize * ctbrows, ctbsize * ctbcols)
    pitch = pitch or W * 4
    dy, du, dv = DeviceBuffer(np.ascontiguousarray(y)), DeviceBuffer(np.ascontiguousarray(u)), DeviceBuffer(np.ascontiguousarray(v))
    do = DeviceBuffer(nbytes=n * H * pitch)
    capi.check(L.ffhip_memset(do.ptr, 0, do.nbytes, None))
    capi.check(L.ffhip_yuv420_to_bgra_16(do.ptr, pitch, dy.ptr, du.ptr, dv.ptr, W, W // 2, ctbrows, ctbcols, ctbsize,
                                         n, H * W, H * W // 4, H * pitch, None), "ffhip_yuv420_to_bgra_16")
    return do.to_host((n, H, pitch), np.uint8)


def yuv400_to_bgra_16(y, ctbrows, ctbcols, ctbsize, pitch=None):
    """YUV400_to_BGRA32_16bit (utils/colorspace.c:715-742)."""
    L = capi.require_device()
    n, H, W = y.shape
    pitch = pitch or W * 4
    dy = DeviceBuffer(np.ascontiguousarray(y))
    do = DeviceBuffer(nbytes=n * H * pitch)
    capi.check(L.ffhip_memset(do.ptr, 0, do.nbytes, None))
    capi.check(L.ffhip_yuv400_to_bgra_16(do.ptr, pitch, dy.ptr, W, ctbrows, ctbcols, ctbsize, n, H * W, H * pitch,
                                         None), "ffhip_yuv400_to_bgra_16")
    return do.to_host((n, H, pitch), np.uint8)


def vp8_residual_batch(levels, mbinfo, quant):
    """Per-macroblock residual of vp8_decode_residual_block (format/webp.c:1147-1196):
    levels int16 [n][25][16], mbinfo uint8 [n][32], quant uint16 [4][8] -> int16 [n][384]."""
    L = capi.require_device()
    n = levels.shape[0]
    assert levels.shape == (n, 25, 16) and mbinfo.shape == (n, 32) and quant.shape == (4, 8)
    dl, di, dq = DeviceBuffer(np.ascontiguousarray(levels)), DeviceBuffer(np.ascontiguousarray(mbinfo)), \
        DeviceBuffer(np.ascontiguousarray(quant))
    do = DeviceBuffer(nbytes=n * 384 * 2)
    capi.check(L.ffhip_vp8_residual_batch(n, dl.ptr, di.ptr, dq.ptr, do.ptr, None), "ffhip_vp8_residual_batch")
    return do.to_host((n, 384), np.int16)


def hevc_residual_batch(n, level, tuinfo, bitdepth=8, epp=False, scaling=None):
    """scale_and_transform (coding/hevc.c:4172-4251) for a batch of n x n TUs:
    level int16 [n_tu][n*n] row-major, tuinfo uint8 [n_tu][4] -> residual int16 [n_tu][n*n]."""
    L = capi.require_device()
    n_tu = level.shape[0]
    assert level.shape == (n_tu, n * n) and tuinfo.shape == (n_tu, 4)
    dl, di = DeviceBuffer(np.ascontiguousarray(level)), DeviceBuffer(np.ascontiguousarray(tuinfo))
    ds = DeviceBuffer(np.ascontiguousarray(scaling)) if scaling is not None else None
    do = DeviceBuffer(nbytes=max(n_tu * n * n * 2, 16))
    capi.check(L.ffhip_hevc_residual_batch(n, n_tu, dl.ptr, di.ptr, ds.ptr if ds else None, bitdepth, int(epp),
                                           do.ptr, None), "ffhip_hevc_residual_batch")
    return do.to_host((n_tu, n * n), np.int16)


def vp8_predict_recon(mbcols, mbrows, modes, residual, resmap=None):
    """vp8_prerdict_mb over whole key frames (format/webp.c:1833-1851, format/predict.c:426-645).
    modes uint8 [n][n_mb][20], residual int16 [n][rows][384], resmap int32 [n][n_mb] or None.
    Returns zero-initialised planes after reconstruction: (Y [n][16r][16c], U, V)."""
    L = capi.require_device()
    n, n_mb = modes.shape[0], mbcols * mbrows
    assert modes.shape == (n, n_mb, 20) and residual.shape[0] == n and residual.shape[2] == 384
    modes = np.ascontiguousarray(modes)
    dm, dr = DeviceBuffer(modes), DeviceBuffer(np.ascontiguousarray(residual))
    dmap = DeviceBuffer(np.ascontiguousarray(resmap, dtype=np.int32)) if resmap is not None else None
    ysz, csz = 256 * n_mb, 64 * n_mb
    dy, du, dv = DeviceBuffer(nbytes=n * ysz), DeviceBuffer(nbytes=n * csz), DeviceBuffer(nbytes=n * csz)
    for d in (dy, du, dv):
        capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
    capi.check(L.ffhip_vp8_predict_recon(mbcols, mbrows, n, modes.ctypes.data, dm.ptr, dr.ptr, residual.shape[1] * 384,
                                         dmap.ptr if dmap else None, dy.ptr, du.ptr, dv.ptr, ysz, csz, None),
               "ffhip_vp8_predict_recon")
    return (dy.to_host((n, 16 * mbrows, 16 * mbcols), np.uint8), du.to_host((n, 8 * mbrows, 8 * mbcols), np.uint8),
            dv.to_host((n, 8 * mbrows, 8 * mbcols), np.uint8))


last_sync_status = 0     # what the stream sync of the last vp8_predict_loopfilter / vp8_decode_frames call returned (0 or capi.FFHIP_RETRIED)


def vp8_predict_loopfilter(mbcols, mbrows, modes, residual, filter_type, filters, resmap=None):
    """ffhip_vp8_predict_loopfilter: prediction + reconstruction and the loop filter of whole key frames as one call
    (format/webp.c:1833-1866), the two row kernels side by side.  Arguments as vp8_predict_recon / vp8_loopfilter;
    returns the filtered planes (Y [n][16r][16c], U, V)."""
    L = capi.require_device()
    n, n_mb = modes.shape[0], mbcols * mbrows
    assert modes.shape == (n, n_mb, 20) and residual.shape[0] == n and residual.shape[2] == 384
    modes = np.ascontiguousarray(modes)
    dm, dr, df = DeviceBuffer(modes), DeviceBuffer(np.ascontiguousarray(residual)), DeviceBuffer(np.ascontiguousarray(filters))
    dmap = DeviceBuffer(np.ascontiguousarray(resmap, dtype=np.int32)) if resmap is not None else None
    ysz, csz = 256 * n_mb, 64 * n_mb
    dy, du, dv = DeviceBuffer(nbytes=n * ysz), DeviceBuffer(nbytes=n * csz), DeviceBuffer(nbytes=n * csz)
    for d in (dy, du, dv):
        capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
    capi.check(L.ffhip_vp8_predict_loopfilter(mbcols, mbrows, n, modes.ctypes.data, dm.ptr, dr.ptr, residual.shape[1] * 384,
                                              dmap.ptr if dmap else None, filter_type, df.ptr, dy.ptr, du.ptr, dv.ptr, ysz, csz, None),
               "ffhip_vp8_predict_loopfilter")
    global last_sync_status
    last_sync_status = capi.sync(None)
    return (dy.to_host((n, 16 * mbrows, 16 * mbcols), np.uint8), du.to_host((n, 8 * mbrows, 8 * mbcols), np.uint8),
            dv.to_host((n, 8 * mbrows, 8 * mbcols), np.uint8))


def vp8_decode_frames(mbcols, mbrows, modes, residual, filter_type, filters, resmap=None, planes=False, pitch=None, host_modes=True):
    """ffhip_vp8_decode_frames: the frame loop of vp8_decode (format/webp.c:1833-1868) for a batch -- prediction, loop filter and
    colour conversion as one call.  Returns BGRA uint8 [n][16r][pitch] and, with planes=True, the filtered (Y, U, V) too."""
    L = capi.require_device()
    n, n_mb = modes.shape[0], mbcols * mbrows
    assert modes.shape == (n, n_mb, 20) and residual.shape[0] == n and residual.shape[2] == 384
    modes = np.ascontiguousarray(modes)
    dm, dr = DeviceBuffer(modes), DeviceBuffer(np.ascontiguousarray(residual))
    df = DeviceBuffer(np.ascontiguousarray(filters)) if filters is not None else None
    dmap = DeviceBuffer(np.ascontiguousarray(resmap, dtype=np.int32)) if resmap is not None else None
    H, W = 16 * mbrows, 16 * mbcols
    pitch = pitch or W * 4
    do = DeviceBuffer(nbytes=n * H * pitch)
    capi.check(L.ffhip_memset(do.ptr, 0, do.nbytes, None))
    ysz, csz = 256 * n_mb, 64 * n_mb
    dy = du = dv = None
    if planes:
        dy, du, dv = DeviceBuffer(nbytes=n * ysz), DeviceBuffer(nbytes=n * csz), DeviceBuffer(nbytes=n * csz)
        for d in (dy, du, dv):
            capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
    capi.check(L.ffhip_vp8_decode_frames(mbcols, mbrows, n, modes.ctypes.data if host_modes else None, dm.ptr, dr.ptr, residual.shape[1] * 384,
                                         dmap.ptr if dmap else None, filter_type, df.ptr if df else None, do.ptr, pitch, H * pitch,
                                         dy.ptr if planes else None, du.ptr if planes else None, dv.ptr if planes else None, ysz, csz, None),
               "ffhip_vp8_decode_frames")
    global last_sync_status
    last_sync_status = capi.sync(None)
    bgra = do.to_host((n, H, pitch), np.uint8)
    if not planes:
        return bgra
    return bgra, (dy.to_host((n, H, W), np.uint8), du.to_host((n, H // 2, W // 2), np.uint8), dv.to_host((n, H // 2, W // 2), np.uint8))


def hevc_decode_tiles(tus, residual, width, height, tile_first=None, bd=8, pitch=None):
    """ffhip_hevc_decode_tiles on a 4:2:0 plane set: intra reconstruction of the (concatenated) TU lists and the colour conversion as one call.
    Returns (bgra uint8 [height][pitch], (Y, U, V) int16 planes)."""
    L = capi.require_device()
    tus = np.ascontiguousarray(tus)
    dt, dr = DeviceBuffer(tus.view(np.uint8)), DeviceBuffer(np.ascontiguousarray(residual))
    cw, ch = width // 2, height // 2
    dy, du, dv = DeviceBuffer(nbytes=width * height * 2), DeviceBuffer(nbytes=cw * ch * 2), DeviceBuffer(nbytes=cw * ch * 2)
    pitch = pitch or width * 4
    do = DeviceBuffer(nbytes=height * pitch)
    for d in (dy, du, dv, do):
        capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
    tf = np.ascontiguousarray(tile_first if tile_first is not None else [0], dtype=np.int64)
    capi.check(L.ffhip_hevc_decode_tiles(tus.ctypes.data, dt.ptr, len(tus), tf.ctypes.data, len(tf), dr.ptr, dy.ptr, du.ptr, dv.ptr, width, height, width, cw, ch, cw, bd, bd,
                                         do.ptr, pitch, None), "ffhip_hevc_decode_tiles")
    capi.check(L.ffhip_stream_sync(None), "ffhip_stream_sync")
    return do.to_host((height, pitch), np.uint8), (dy.to_host((height, width), np.int16), du.to_host((ch, cw), np.int16), dv.to_host((ch, cw), np.int16))


def hevc_intra_recon(tus, residual, width, height, chroma=True, bd_y=8, bd_c=8, csub=2):
    """decode_intra_block steps 5-10 (coding/hevc.c:4730-4790) for a TU list in decode order
    (structured array of dtype synth.HEVC_TU_DTYPE == struct ffhip_hevc_tu); planes start at 0;
    csub = chroma subsampling divisor (2 for 4:2:0, 1 for 4:4:4)."""
    L = capi.require_device()
    tus = np.ascontiguousarray(tus)
    assert tus.dtype.itemsize == 32
    dt, dr = DeviceBuffer(tus.view(np.uint8)), DeviceBuffer(np.ascontiguousarray(residual))
    cw, ch = (width // csub, height // csub) if chroma else (0, 0)
    dy = DeviceBuffer(nbytes=width * height * 2)
    du = DeviceBuffer(nbytes=max(cw * ch * 2, 16))
    dv = DeviceBuffer(nbytes=max(cw * ch * 2, 16))
    for d in (dy, du, dv):
        capi.check(L.ffhip_memset(d.ptr, 0, d.nbytes, None))
    capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.ptr, len(tus), dr.ptr, dy.ptr, du.ptr if chroma else None,
                                        dv.ptr if chroma else None, width, height, width, cw, ch, cw, bd_y, bd_c, None),
               "ffhip_hevc_intra_recon")
    if chroma:
        return dy.to_host((height, width), np.int16), du.to_host((ch, cw), np.int16), dv.to_host((ch, cw), np.int16)
    return dy.to_host((height, width), np.int16), None, None


def jpeg_probe(data):
    """Geometry of a baseline JPEG file (bytes) -> (JpegGeom, width, height)."""
    L = capi.lib()
    g, w, h = capi.JpegGeom(), C.c_int(), C.c_int()
    buf = np.frombuffer(data, dtype=np.uint8)
    capi.check(L.ffhip_jpeg_probe(buf.ctypes.data, buf.size, C.byref(g), C.byref(w), C.byref(h)), "ffhip_jpeg_probe")
    return g, w.value, h.value


def jpeg_entropy_batch(files, n_threads=4):
    """Host-side Huffman decode of same-geometry JPEG files (list of bytes) into the planes the
    reconstruction reads (format/jpg.c:255-415, 588-655).  Returns (geom, cy, cu, cv, quant[n][4][64])."""
    L = capi.lib()
    g, _, _ = jpeg_probe(files[0])
    n = len(files)
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    cy = np.empty(n * g.y_blocks * 64, np.int16)
    cu = np.empty(n * g.c_blocks * 64, np.int16) if g.ncomp == 3 else None
    cv = np.empty(n * g.c_blocks * 64, np.int16) if g.ncomp == 3 else None
    quant = np.empty((n, 4, 64), np.uint16)
    status = (C.c_int * n)()
    capi.check(L.ffhip_jpeg_entropy_batch(ptrs, lens, n, n_threads, C.byref(g), _vp(cy), _vp(cu), _vp(cv), _vp(quant), status),
               "ffhip_jpeg_entropy_batch")
    return g, cy, cu, cv, quant


def decode_jpeg_files(files, n_threads=4):
    """transbmp-equivalent core: files -> BGRA [n][H][W][4] (coded size), entropy on the host
    threads, reconstruction on the GPU."""
    g, cy, cu, cv, quant = jpeg_entropy_batch(files, n_threads)
    return g, jpeg_recon_batch_host(g, len(files), cy, cu, cv, quant)


def vp8_loopfilter(mbcols, mbrows, filter_type, modes, filters, y, u, v):
    """loopfilter over whole key frames (format/webp.c:1686-1752, 1856-1866); planes uint8
    [n][H][W] are filtered and returned."""
    L = capi.require_device()
    n = modes.shape[0]
    dm, df = DeviceBuffer(np.ascontiguousarray(modes)), DeviceBuffer(np.ascontiguousarray(filters))
    dy, du, dv = DeviceBuffer(np.ascontiguousarray(y)), DeviceBuffer(np.ascontiguousarray(u)), DeviceBuffer(np.ascontiguousarray(v))
    n_mb = mbcols * mbrows
    capi.check(L.ffhip_vp8_loopfilter(mbcols, mbrows, n, filter_type, dm.ptr, df.ptr, dy.ptr, du.ptr, dv.ptr,
                                      256 * n_mb, 64 * n_mb, None), "ffhip_vp8_loopfilter")
    return dy.to_host(y.shape, np.uint8), du.to_host(u.shape, np.uint8), dv.to_host(v.shape, np.uint8)


def heif_grid_parse(item):
    """ImageGrid item payload (bytes) -> capi.HeifGrid, as decode_grid_items reads it (format/heif.c:273-298)."""
    L = capi.lib()
    g = capi.HeifGrid()
    buf = (C.c_uint8 * len(item)).from_buffer_copy(bytes(item))
    capi.check(L.ffhip_heif_grid_parse(C.cast(buf, C.c_void_p), len(item), C.byref(g)), "ffhip_heif_grid_parse")
    return g


def heif_grid_compose(tiles, cols, out_w, out_h):
    """tiles uint8 [rows*cols][tile_h][tile_w][4] (row-major `dimg` order) -> canvas [out_h][out_w][4]."""
    L = capi.require_device()
    tiles = np.ascontiguousarray(tiles)
    n, th, tw, _ = tiles.shape
    dt = DeviceBuffer(tiles)
    dc = DeviceBuffer(nbytes=out_w * out_h * 4)
    capi.check(L.ffhip_heif_grid_compose(dc.ptr, out_w * 4, out_w, out_h, dt.ptr, tw * 4, tw * th * 4, tw, th, n // cols, cols,
                                         None), "ffhip_heif_grid_compose")
    return dc.to_host((out_h, out_w, 4), np.uint8)


class PinnedArray:
    """numpy view of pinned host memory (ffhip_host_malloc); keep the object alive while the view is used."""

    def __init__(self, shape, dtype=np.uint8):
        L = capi.require_device()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = L.ffhip_host_malloc(self.nbytes)
        if not self.ptr:
            raise capi.FfhipError("ffhip_host_malloc failed")
        self.array = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.ptr), dtype=dtype).reshape(shape)

    def __del__(self):
        try:
            capi.lib().ffhip_host_free(self.ptr)
        except Exception:
            pass


def jpeg_decode_files(files, n_threads=8, chunk=0, out=None):
    """ffhip_jpeg_decode_files: same-geometry baseline JPEG files (list of bytes) -> (geom, BGRA [n][H][W][4] at the
    coded size), entropy decode on host threads overlapped with copy + reconstruction on the GPU.  `out`: a
    preallocated uint8 array [n][H][W][4] (e.g. PinnedArray(...).array, which takes the device copy directly)."""
    L = capi.require_device()
    g, _, _ = jpeg_probe(files[0])
    n = len(files)
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    if out is None:
        out = np.empty((n, g.height, g.width, 4), np.uint8)
    assert out.shape == (n, g.height, g.width, 4) and out.dtype == np.uint8 and out.flags.c_contiguous
    status = (C.c_int * n)()
    g2 = capi.JpegGeom()
    capi.check(L.ffhip_jpeg_decode_files(ptrs, lens, n, n_threads, chunk, C.byref(g2), out.ctypes.data, g.width * 4,
                                         g.width * 4 * g.height, status), "ffhip_jpeg_decode_files")
    return g2, out


def jpeg_entropy_batch_gpu(files, n_threads=4):
    """ffhip_jpeg_entropy_batch_gpu on files with restart markers; planes come back to the host for inspection.
    Returns (geom, cy, cu, cv, quant[n][4][64]) like jpeg_entropy_batch."""
    L = capi.require_device()
    g, _, _ = jpeg_probe(files[0])
    n = len(files)
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    dy = DeviceBuffer(nbytes=n * g.y_blocks * 128)
    du = DeviceBuffer(nbytes=max(n * g.c_blocks * 128, 16)) if g.ncomp == 3 else None
    dv = DeviceBuffer(nbytes=max(n * g.c_blocks * 128, 16)) if g.ncomp == 3 else None
    dq = DeviceBuffer(nbytes=n * 512)
    status = (C.c_int * n)()
    capi.check(L.ffhip_jpeg_entropy_batch_gpu(ptrs, lens, n, n_threads, C.byref(g), dy.ptr, du.ptr if du else None, dv.ptr if dv else None, dq.ptr,
                                              status, None), "ffhip_jpeg_entropy_batch_gpu")
    cy = dy.to_host((n * g.y_blocks * 64,), np.int16)
    cu = du.to_host((n * g.c_blocks * 64,), np.int16) if du else None
    cv = dv.to_host((n * g.c_blocks * 64,), np.int16) if dv else None
    return g, cy, cu, cv, dq.to_host((n, 4, 64), np.uint16)


def jpeg_decode_files_device(files, n_threads=8):
    """ffhip_jpeg_decode_files_device: files -> BGRA left in device memory; returned here as a host copy for checks,
    together with the DeviceBuffer that holds it: (geom, host array [n][H][W][4], device buffer)."""
    L = capi.require_device()
    g, _, _ = jpeg_probe(files[0])
    n = len(files)
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    dout = DeviceBuffer(nbytes=n * g.width * g.height * 4)
    status = (C.c_int * n)()
    g2 = capi.JpegGeom()
    capi.check(L.ffhip_jpeg_decode_files_device(ptrs, lens, n, n_threads, C.byref(g2), dout.ptr, g.width * 4, g.width * 4 * g.height,
                                                status, None), "ffhip_jpeg_decode_files_device")
    return g2, dout.to_host((n, g.height, g.width, 4), np.uint8), dout
