"""CPU: the C-ABI library builds, loads without a GPU, exports every symbol that
include/ffpic_hip.h declares, keeps the reference's struct layouts, and refuses to
compute (loudly) when no gfx950 device exists -- there is no CPU fallback."""
import ctypes as C
import os
import re

import pytest

from ffpic_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.lib()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ffpic_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(ffhip_[a-z0-9_]+|hip_accl_[a-z]+)\s*\(", text))
    return names - {"ffhip_accl_ops", "ffhip_dct_ops", "ffhip_cs_ops"}


def test_header_and_exports_agree(L):
    decl = declared_functions()
    assert decl == set(capi.EXPORTS), (decl ^ set(capi.EXPORTS))
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/ffpic_hip.h but not exported"


def test_struct_layouts_match_reference():
    # struct accl_ops (arch/accl.h:20-25) on LP64: fn ptrs @0,@8; type @16; TAILQ @24,@32; sizeof 40
    assert C.sizeof(capi.AcclOps) == 40
    assert capi.AcclOps.idct_4x4.offset == 0 and capi.AcclOps.idct_8x8.offset == 8
    assert capi.AcclOps.type.offset == 16 and capi.AcclOps.tqe_next.offset == 24 and capi.AcclOps.tqe_prev.offset == 32
    # struct dct_ops (utils/idct.h:14-21): int @0, four fn ptrs @8..
    assert C.sizeof(capi.DctOps) == 40 and capi.DctOps.idct_4x4.offset == 8 and capi.DctOps.idct_8x8.offset == 16
    # struct cs_ops (utils/colorspace.h:29-33)
    assert C.sizeof(capi.CsOps) == 16
    assert C.sizeof(capi.JpegGeom) == 32


def test_no_silent_cpu_fallback(L):
    if L.ffhip_device_count() > 0:
        pytest.skip("a GPU is present; covered by the -m gpu tests")
    with pytest.raises(capi.FfhipError):
        capi.require_device()
    g = capi.jpeg_geom(4, 2)
    # argument validation comes first, then the device check: never a computed result
    assert L.ffhip_jpeg_recon_batch(C.byref(g), 1, None, None, None, None, 0, None, 0, 0, None, 0, None) == -22
    assert not L.ffhip_accl_ops_get()          # back-end that cannot run exposes no ops ...
    assert not L.ffhip_get_dct_ops(16) and not L.ffhip_get_cs_ops(16)
    L.hip_accl_init()                          # ... and does not register (arch/opencl/opcl.c:112-114)
    assert L.ffhip_malloc(1024) is None
    assert L.ffhip_jpeg_recon_batch_host(C.byref(g), 1, None, None, None, None, 0, None, 0, 0) == -19
    # every later entry point refuses the same way: no device, no result
    import os
    import numpy as np
    from conftest import GOLDEN
    data = np.frombuffer(open(os.path.join(GOLDEN, "file_q80_grey.jpg"), "rb").read(), np.uint8)
    ptrs, lens, status = (C.c_void_p * 1)(data.ctypes.data), (C.c_size_t * 1)(data.size), (C.c_int * 1)()
    out = np.zeros(160 * 96 * 4, np.uint8)
    assert L.ffhip_jpeg_decode_files(ptrs, lens, 1, 2, 0, None, out.ctypes.data, 160 * 4, 160 * 96 * 4, status) == -19
    assert not out.any()
    assert L.ffhip_host_malloc(64) is None
    buf = np.zeros(4096, np.uint8)
    p = buf.ctypes.data
    assert L.ffhip_heif_grid_compose(p, 64, 16, 16, p, 64, 1024, 16, 16, 1, 1, None) == -19
    assert L.ffhip_yuv420_to_bgra(p, 64, p, p, p, 16, 8, 1, 1, 1, 0, 0, 0, None) == -19
    assert L.ffhip_vp8_residual_batch(1, p, p, p, p, None) == -19
    assert L.ffhip_hevc_residual_batch(4, 1, p, p, None, 8, 0, p, None) == -19
    modes = np.zeros(20, np.uint8)
    assert L.ffhip_vp8_predict_recon(1, 1, 1, modes.ctypes.data, p, p, 384, None, p, p, p, 256, 64, None) == -19
    assert L.ffhip_vp8_loopfilter(1, 1, 1, 2, p, p, p, p, p, 256, 64, None) == -19
    assert L.ffhip_vp8_predict_loopfilter(1, 1, 1, modes.ctypes.data, p, p, 384, None, 2, p, p, p, p, 256, 64, None) == -19
    assert L.ffhip_vp8_predict_loopfilter(1, 1, 1, modes.ctypes.data, p, p, 384, None, 3, p, p, p, p, 256, 64, None) == -22   # no such filter type
    assert L.ffhip_vp8_predict_loopfilter(1, 1, 1, modes.ctypes.data, p, p, 384, None, 2, None, p, p, p, 256, 64, None) == -22  # a filter without parameters
    tu = np.zeros(32, np.uint8)
    tu[4] = 2                                   # log2_size
    assert L.ffhip_hevc_intra_recon(tu.ctypes.data, p, 1, p, p, None, None, 16, 16, 16, 0, 0, 0, 8, 8, None) == -19
    assert not buf.any()


def test_geometry_validation(L):
    for bad in (capi.jpeg_geom(0, 2), capi.jpeg_geom(4, 2, ncomp=2), capi.jpeg_geom(4, 2, h=3),
                capi.jpeg_geom(4, 2, qt_id=(0, 4, 1))):
        assert L.ffhip_jpeg_recon_batch(C.byref(bad), 1, 16, 16, 16, 16, 0, 16, 4096, 0, None, 0, None) == -22
        assert L.ffhip_jpeg_workspace_bytes(C.byref(bad), 1) == 0
    g = capi.jpeg_geom(4, 2)
    assert L.ffhip_jpeg_recon_batch(C.byref(g), 0, None, None, None, None, 0, None, 0, 0, None, 0, None) == 0  # empty batch
    assert L.ffhip_jpeg_workspace_bytes(C.byref(g), 8) == 0          # fused path needs none
    g444 = capi.jpeg_geom(4, 2, h=1, v=1)
    assert L.ffhip_jpeg_workspace_bytes(C.byref(g444), 2) == 0       # 4:4:4 / 4:2:2 / 4:4:0 / grey: fused strip kernel
    assert L.ffhip_jpeg_kernel_name(C.byref(g444)) == b"k_jpeg_fused_strip"
    ggrey4 = capi.jpeg_geom(4, 2, ncomp=1, h=2, v=2)                  # the one layout left on the two-pass path
    assert L.ffhip_jpeg_workspace_bytes(C.byref(ggrey4), 2) == 2 * 8 * 4 * 128
    assert L.ffhip_jpeg_kernel_name(C.byref(g)) == b"k_jpeg420_fused"


def test_product_never_touches_the_oracle():
    """The product tree must not import, link or reference the CPU checker."""
    pkg = os.path.join(ROOT, "ffpic_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text and "libffo" not in text and "ffo_" not in text, os.path.join(dirpath, f)


def test_switch_values_are_kept_whole_and_survive_reloads(L, monkeypatch):
    """FFHIP_* values are interned, not cut at a fixed width (FFHIP_RCCL_LIB is a path: a torch wheel's librccl.so is 60
    characters away), a pointer handed out stays valid across ffhip_reload_env, and lookups may race with reloads."""
    import threading
    path = "/usr/local/lib/python3.10/dist-packages/torch/lib/" + "x" * 150 + "/librccl.so"
    buf = C.create_string_buffer(512)
    monkeypatch.setenv("FFHIP_TEST_SWITCH", path)
    capi.reload_env()
    assert L.ffhip_env_value_test(b"FFHIP_TEST_SWITCH", buf, 512) == len(path) and buf.value.decode() == path
    monkeypatch.setenv("FFHIP_TEST_SWITCH", "1")
    assert L.ffhip_env_value_test(b"FFHIP_TEST_SWITCH", buf, 512) == len(path)          # read once per process ...
    capi.reload_env()
    assert L.ffhip_env_value_test(b"FFHIP_TEST_SWITCH", buf, 512) == 1 and buf.value == b"1"   # ... until told otherwise
    monkeypatch.delenv("FFHIP_TEST_SWITCH")
    capi.reload_env()
    assert L.ffhip_env_value_test(b"FFHIP_TEST_SWITCH", buf, 512) == -1
    # lookups from four threads while the main thread flips the value and reloads: every answer is one of the values set
    values = [b"a" * 10, b"b" * 100, b"c" * 300]
    bad, stop = [], threading.Event()

    def reader():
        b = C.create_string_buffer(512)
        while not stop.is_set():
            n = L.ffhip_env_value_test(b"FFHIP_TEST_SWITCH2", b, 512)
            if n >= 0 and b.value not in values:
                bad.append(b.value)
    threads = [threading.Thread(target=reader) for _ in range(4)]
    for t in threads:
        t.start()
    for i in range(300):
        os.environ["FFHIP_TEST_SWITCH2"] = values[i % 3].decode()
        capi.reload_env()
    stop.set()
    for t in threads:
        t.join()
    os.environ.pop("FFHIP_TEST_SWITCH2", None)
    capi.reload_env()
    assert not bad


def test_bgra_layout_recommendation(L):
    """ffhip_bgra_layout: the reference's pitch plus one KiB, stride = pitch x coded height; needs no device"""
    g = capi.jpeg_geom(240, 135)
    p, s = C.c_int64(), C.c_int64()
    assert L.ffhip_bgra_layout(C.byref(g), C.byref(p), C.byref(s)) == 0
    assert (p.value, s.value) == (3840 * 4 + 1024, (3840 * 4 + 1024) * 2160)
    g = capi.jpeg_geom(10, 7, 3, 4, 1, (0, 1, 1))
    assert L.ffhip_bgra_layout(C.byref(g), C.byref(p), C.byref(s)) == 0 and p.value == 320 * 4 + 1024 and s.value == p.value * 56
    assert L.ffhip_bgra_layout(C.byref(g), None, C.byref(s)) == capi.FFHIP_EINVAL
