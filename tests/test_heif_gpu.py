"""HEIF image grid: item parsing on the host, tile placement on the GPU.

The reference never places tiles (format/heif.c:305 decodes them all into one buffer), so the
expected canvas is the ISO/IEC 23008-12 6.6.2.3.1 definition restated in numpy below; parity for
this row is unpinned."""
import struct

import numpy as np
import pytest

from ffpic_amd import capi, ops


def expected_canvas(tiles, cols, out_w, out_h):
    n, th, tw, _ = tiles.shape
    rows = n // cols
    full = tiles.reshape(rows, cols, th, tw, 4).transpose(0, 2, 1, 3, 4).reshape(rows * th, cols * tw, 4)
    return full[:out_h, :out_w]


def test_grid_item_parse():
    """the two item layouts decode_grid_items accepts (format/heif.c:284-296) and the lengths it asserts"""
    g = ops.heif_grid_parse(bytes([0, 0, 2, 3]) + struct.pack(">HH", 4032, 3024))
    assert (g.rows, g.cols, g.output_width, g.output_height) == (3, 4, 4032, 3024)
    g = ops.heif_grid_parse(bytes([0, 1, 0, 0]) + struct.pack(">II", 70000, 66000))
    assert (g.rows, g.cols, g.output_width, g.output_height, g.flags) == (1, 1, 70000, 66000, 1)
    for bad in (bytes([0, 0, 1, 1]) + b"\0" * 8, bytes([0, 1, 1, 1]) + b"\0" * 4, b"\0\0\0", bytes(8)):
        with pytest.raises(capi.FfhipError):
            ops.heif_grid_parse(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,tw,th,ow,oh", [(2, 3, 64, 32, 180, 50), (1, 1, 16, 16, 16, 16), (3, 2, 30, 18, 47, 40),
                                                   (2, 2, 512, 512, 1000, 1023), (4, 5, 20, 12, 81, 37), (1, 4, 8, 8, 29, 8)])
def test_grid_compose(rows, cols, tw, th, ow, oh):
    rng = np.random.default_rng(rows * 100 + cols)
    tiles = rng.integers(0, 256, size=(rows * cols, th, tw, 4), dtype=np.uint8)
    got = ops.heif_grid_compose(tiles, cols, ow, oh)
    assert np.array_equal(got, expected_canvas(tiles, cols, ow, oh))


@pytest.mark.gpu
def test_grid_compose_rejects_bad_geometry():
    L = capi.require_device()
    d = ops.DeviceBuffer(nbytes=1 << 16)
    # tiles do not cover the canvas / a whole tile row lies outside it / pitch too small / misaligned
    for args in ((d.ptr, 400, 100, 64, d.ptr, 128, 128 * 32, 32, 32, 2, 2),
                 (d.ptr, 256, 64, 30, d.ptr, 128, 128 * 32, 32, 32, 2, 2),
                 (d.ptr, 100, 64, 64, d.ptr, 128, 128 * 32, 32, 32, 2, 2),
                 (d.ptr + 2, 256, 64, 64, d.ptr, 128, 128 * 32, 32, 32, 2, 2)):
        assert L.ffhip_heif_grid_compose(*args, None) == capi.FFHIP_EINVAL


def test_hevc_picture_layout():
    """the buffer arithmetic of parse_slice_segment_layer (coding/hevc.c:7223-7236, 7258-7277)"""
    import ctypes as C
    L = capi.lib()
    lay = capi.HevcLayout()
    for (w, h, lg) in ((1920, 1080, 6), (7680, 4320, 6), (1918, 1081, 4), (33, 17, 5)):
        assert L.ffhip_hevc_picture_layout(w, h, lg, C.byref(lay)) == 0
        height = ((h + 3) >> 2) << 2
        ys = ((w + 3) >> 2) << 2
        assert (lay.height, lay.y_stride, lay.uv_stride) == (height, ys, ys >> 1)
        assert (lay.size, lay.u_offset, lay.v_offset) == (height * ys, height * ys, height * ys * 3 // 2)
        assert lay.pitch == ((ys * 32 + 31) >> 5) << 2
        assert (lay.ctbrows, lay.ctbcols) == (-(-height // (1 << lg)), -(-w // (1 << lg)))
    assert L.ffhip_hevc_picture_layout(0, 10, 6, C.byref(lay)) == capi.FFHIP_EINVAL
    assert L.ffhip_hevc_picture_layout(64, 64, 7, C.byref(lay)) == capi.FFHIP_EINVAL


@pytest.mark.gpu
def test_grid_item_of_the_reference_decoded_heic(golden):
    """the ImageGrid payload of tests/golden/file_f_grid.heic -- a file the reference's own HEIF loader decoded through
    decode_grid_items -- parsed by ffhip_heif_grid_parse, and its one tile composed by ffhip_heif_grid_compose: the canvas is
    the reference's picture"""
    g = golden("hevc_file.npz")
    w, h, _ = [int(x) for x in g["f_dims"]]
    grid = ops.heif_grid_parse(bytes(g["f_grid"]))
    assert (grid.rows, grid.cols, grid.output_width, grid.output_height) == (1, 1, w, h)
    tile = np.ascontiguousarray(g["f_bgra"]).reshape(1, h, w, 4)
    got = ops.heif_grid_compose(tile, grid.cols, grid.output_width, grid.output_height)
    assert np.array_equal(np.asarray(got).reshape(h, w * 4), g["f_bgra"])
