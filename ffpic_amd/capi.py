"""ctypes binding of libffpic_hip.so (include/ffpic_hip.h).  Loading never touches
the GPU; compute entry points fail loudly (FfhipError) without a gfx950 device."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, os.environ.get("FFHIP_LIB") or "libffpic_hip.so")   # FFHIP_LIB: A/B runs of two builds in one gpurun call (tests/tools/ab_*.sh)
CSRC = os.path.join(HERE, "csrc")

# every symbol include/ffpic_hip.h declares (tests check the library exports them all)
EXPORTS = [
    "ffhip_device_count", "ffhip_init", "ffhip_shutdown", "ffhip_strerror", "ffhip_arch_name",
    "ffhip_malloc", "ffhip_free", "ffhip_memcpy_h2d", "ffhip_memcpy_d2h", "ffhip_memset",
    "ffhip_stream_create", "ffhip_stream_destroy", "ffhip_stream_sync",
    "ffhip_event_create", "ffhip_event_destroy", "ffhip_event_record", "ffhip_event_elapsed_ms",
    "hip_accl_init", "hip_accl_uninit", "ffhip_accl_ops_get",
    "ffhip_get_dct_ops", "ffhip_get_cs_ops", "ffhip_idct_4x4_hevc",
    "ffhip_jpeg_recon_batch", "ffhip_jpeg_workspace_bytes", "ffhip_jpeg_recon_batch_host",
    "ffhip_jpeg_kernel_name", "ffhip_copy_calibrate", "ffhip_jpeg_pattern_calibrate",
    "ffhip_yuv420_to_bgra", "ffhip_yuv420_to_bgra_16", "ffhip_yuv400_to_bgra_16",
    "ffhip_vp8_residual_batch", "ffhip_hevc_residual_batch", "ffhip_vp8_predict_recon",
    "ffhip_hevc_intra_recon", "ffhip_hevc_intra_plan", "ffhip_debug_hevc_plan_result", "ffhip_vp8_decode_frames_form", "ffhip_debug_huff_times", "ffhip_hevc_intra_recon_tiles", "ffhip_hevc_decode_tiles", "ffhip_vp8_loopfilter",
    "ffhip_jpeg_probe", "ffhip_jpeg_entropy_decode", "ffhip_jpeg_entropy_decode_mt", "ffhip_jpeg_entropy_batch", "ffhip_bmp_write",
    "ffhip_heif_grid_parse", "ffhip_heif_grid_compose", "ffhip_hevc_picture_layout", "ffhip_jpeg_decode_files", "ffhip_jpeg_decode_files_device", "ffhip_jpeg_entropy_batch_gpu", "ffhip_jpeg_stage_scan_test", "ffhip_jpeg_stage_scan_raw_test", "ffhip_jpeg_lut_test", "ffhip_host_malloc", "ffhip_host_free",
    "ffhip_shard_range", "ffhip_comm_unique_id", "ffhip_comm_init_rank", "ffhip_comm_destroy", "ffhip_batch_close", "ffhip_batch_complete",
    "ffhip_bgra_checksum", "ffhip_vp8_filter_params", "ffhip_vp8_predict_loopfilter", "ffhip_reload_env", "ffhip_env_value_test", "ffhip_vp8_decode_frames", "ffhip_bgra_layout",
]


FFHIP_EINVAL, FFHIP_ENOMEM, FFHIP_ENODEV, FFHIP_EIO = -22, -12, -19, -5     # include/ffpic_hip.h:34-37


class FfhipError(RuntimeError):
    pass


class HeifGrid(C.Structure):
    """ffhip_heif_grid"""
    _fields_ = [("version", C.c_uint8), ("flags", C.c_uint8), ("rows", C.c_uint16), ("cols", C.c_uint16),
                ("output_width", C.c_uint32), ("output_height", C.c_uint32)]


class HevcLayout(C.Structure):
    """ffhip_hevc_layout"""
    _fields_ = [("height", C.c_int32), ("y_stride", C.c_int32), ("uv_stride", C.c_int32), ("size", C.c_int64), ("u_offset", C.c_int64),
                ("v_offset", C.c_int64), ("pitch", C.c_int32), ("ctbrows", C.c_int32), ("ctbcols", C.c_int32)]


class BatchRecord(C.Structure):
    """ffhip_batch_record"""
    _fields_ = [("rank", C.c_int32), ("status", C.c_int32), ("first", C.c_int64), ("count", C.c_int64), ("checksum", C.c_uint64)]


class Vp8FilterHeader(C.Structure):
    """ffhip_vp8_filter_header"""
    _fields_ = [("filter_type", C.c_uint8), ("loop_filter_level", C.c_uint8), ("sharpness_level", C.c_uint8),
                ("segmentation_enabled", C.c_uint8), ("segment_feature_mode", C.c_uint8), ("lf_update_value", C.c_int8 * 4),
                ("loop_filter_adj_enable", C.c_uint8), ("mode_ref_lf_delta0", C.c_int8), ("mb_mode_delta0", C.c_int8),
                ("nbr_partitions", C.c_uint8)]


class JpegGeom(C.Structure):
    """ffhip_jpeg_geom"""
    _fields_ = [("mcu_cols", C.c_int32), ("mcu_rows", C.c_int32), ("ncomp", C.c_int32),
                ("h", C.c_int32), ("v", C.c_int32), ("qt_id", C.c_int32 * 3)]

    @property
    def width(self):
        return self.mcu_cols * 8 * self.h

    @property
    def height(self):
        return self.mcu_rows * 8 * self.v

    @property
    def y_blocks(self):
        return self.mcu_cols * self.mcu_rows * self.h * self.v

    @property
    def c_blocks(self):
        return self.mcu_cols * self.mcu_rows


def jpeg_geom(mcu_cols, mcu_rows, ncomp=3, h=2, v=2, qt_id=(0, 1, 1)):
    g = JpegGeom()
    g.mcu_cols, g.mcu_rows, g.ncomp, g.h, g.v = mcu_cols, mcu_rows, ncomp, h, v
    for i in range(3):
        g.qt_id[i] = qt_id[i]
    return g


class AcclOps(C.Structure):
    """struct ffhip_accl_ops == struct accl_ops (arch/accl.h:20-25)"""
    _fields_ = [("idct_4x4", C.CFUNCTYPE(None, C.c_void_p, C.c_int)),
                ("idct_8x8", C.CFUNCTYPE(None, C.c_void_p, C.c_int)),
                ("type", C.c_int),
                ("tqe_next", C.c_void_p), ("tqe_prev", C.c_void_p)]


class DctOps(C.Structure):
    _fields_ = [("bitdepth", C.c_int),
                ("idct_4x4", C.CFUNCTYPE(None, C.c_void_p, C.c_int)),
                ("idct_8x8", C.CFUNCTYPE(None, C.c_void_p, C.c_int)),
                ("fdct_4x4", C.c_void_p), ("fdct_8x8", C.c_void_p)]


class CsOps(C.Structure):
    _fields_ = [("YUV_to_BGRA32", C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int)),
                ("YUV420_to_BGRA32", C.c_void_p)]


def build(force=False):
    """Compile libffpic_hip.so in-tree with hipcc for gfx950 (works without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", CSRC, "-j4"])


_lib = None


def lib():
    """The loaded library.  Raises FfhipError if it has not been built: there is no
    fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FfhipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(or make -C ffpic_amd/csrc); there is no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i64, sz = C.c_void_p, C.c_int64, C.c_size_t
    L.ffhip_device_count.restype = C.c_int
    L.ffhip_init.argtypes = [C.c_int]
    L.ffhip_strerror.argtypes = [C.c_int]
    L.ffhip_strerror.restype = C.c_char_p
    L.ffhip_arch_name.restype = C.c_char_p
    L.ffhip_malloc.argtypes = [sz]
    L.ffhip_malloc.restype = vp
    L.ffhip_free.argtypes = [vp]
    L.ffhip_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    L.ffhip_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    L.ffhip_memset.argtypes = [vp, C.c_int, sz, vp]
    L.ffhip_stream_create.restype = vp
    L.ffhip_stream_destroy.argtypes = [vp]
    L.ffhip_stream_sync.argtypes = [vp]
    L.ffhip_event_create.restype = vp
    L.ffhip_event_destroy.argtypes = [vp]
    L.ffhip_event_record.argtypes = [vp, vp]
    L.ffhip_event_elapsed_ms.argtypes = [vp, vp]
    L.ffhip_event_elapsed_ms.restype = C.c_float
    L.ffhip_accl_ops_get.restype = C.POINTER(AcclOps)
    L.ffhip_get_dct_ops.argtypes = [C.c_int]
    L.ffhip_get_dct_ops.restype = C.POINTER(DctOps)
    L.ffhip_get_cs_ops.argtypes = [C.c_int]
    L.ffhip_get_cs_ops.restype = C.POINTER(CsOps)
    L.ffhip_idct_4x4_hevc.argtypes = [vp, vp, C.c_int, C.c_bool]
    L.ffhip_idct_4x4_hevc.restype = None
    L.ffhip_jpeg_recon_batch.argtypes = [C.POINTER(JpegGeom), C.c_int, vp, vp, vp, vp, i64, vp, i64, i64, vp, sz, vp]
    L.ffhip_jpeg_workspace_bytes.argtypes = [C.POINTER(JpegGeom), C.c_int]
    L.ffhip_jpeg_workspace_bytes.restype = sz
    L.ffhip_jpeg_recon_batch_host.argtypes = [C.POINTER(JpegGeom), C.c_int, vp, vp, vp, vp, i64, vp, i64, i64]
    L.ffhip_jpeg_kernel_name.argtypes = [C.POINTER(JpegGeom)]
    L.ffhip_jpeg_kernel_name.restype = C.c_char_p
    L.ffhip_copy_calibrate.argtypes = [vp, vp, sz, vp]
    L.ffhip_jpeg_pattern_calibrate.argtypes = [C.POINTER(JpegGeom), C.c_int, vp, vp, vp, vp, i64, vp, i64, i64, vp]
    ci = C.c_int
    L.ffhip_yuv420_to_bgra.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, i64, i64, i64, vp]
    L.ffhip_yuv420_to_bgra_16.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, i64, i64, i64, vp]
    L.ffhip_yuv400_to_bgra_16.argtypes = [vp, ci, vp, ci, ci, ci, ci, ci, i64, i64, vp]
    L.ffhip_vp8_residual_batch.argtypes = [C.c_longlong, vp, vp, vp, vp, vp]
    L.ffhip_hevc_residual_batch.argtypes = [ci, C.c_longlong, vp, vp, vp, ci, ci, vp, vp]
    L.ffhip_jpeg_probe.argtypes = [vp, sz, C.POINTER(JpegGeom), C.POINTER(ci), C.POINTER(ci)]
    L.ffhip_jpeg_entropy_decode.argtypes = [vp, sz, C.POINTER(JpegGeom), vp, vp, vp, vp]
    L.ffhip_jpeg_entropy_decode_mt.argtypes = [vp, sz, C.POINTER(JpegGeom), vp, vp, vp, vp, ci]
    L.ffhip_jpeg_entropy_batch.argtypes = [vp, vp, ci, ci, C.POINTER(JpegGeom), vp, vp, vp, vp, vp]
    L.ffhip_bmp_write.argtypes = [C.c_char_p, vp, ci, ci, i64]
    L.ffhip_host_malloc.argtypes = [sz]
    L.ffhip_host_malloc.restype = vp
    L.ffhip_host_free.argtypes = [vp]
    L.ffhip_host_free.restype = None
    L.ffhip_jpeg_entropy_batch_gpu.argtypes = [vp, vp, ci, ci, C.POINTER(JpegGeom), vp, vp, vp, vp, vp, vp]
    L.ffhip_jpeg_decode_files_device.argtypes = [vp, vp, ci, ci, C.POINTER(JpegGeom), vp, i64, i64, vp, vp]
    L.ffhip_jpeg_decode_files.argtypes = [vp, vp, ci, ci, ci, C.POINTER(JpegGeom), vp, i64, i64, vp]
    L.ffhip_hevc_picture_layout.argtypes = [ci, ci, ci, C.POINTER(HevcLayout)]
    L.ffhip_heif_grid_parse.argtypes = [vp, sz, C.POINTER(HeifGrid)]
    L.ffhip_heif_grid_compose.argtypes = [vp, i64, ci, ci, vp, i64, i64, ci, ci, ci, ci, vp]
    L.ffhip_hevc_intra_plan.argtypes = [vp, C.c_longlong, ci, ci, ci, ci, ci, vp, vp, vp]
    L.ffhip_debug_hevc_plan_result.argtypes = [vp]
    L.ffhip_hevc_intra_recon.argtypes = [vp, vp, C.c_longlong, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.ffhip_hevc_intra_recon_tiles.argtypes = [vp, vp, C.c_longlong, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp]
    L.ffhip_hevc_decode_tiles.argtypes = [vp, vp, C.c_longlong, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, i64, vp]
    L.ffhip_vp8_loopfilter.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, i64, i64, vp]
    L.ffhip_vp8_predict_recon.argtypes = [ci, ci, ci, vp, vp, vp, i64, vp, vp, vp, vp, i64, i64, vp]
    L.ffhip_vp8_predict_loopfilter.argtypes = [ci, ci, ci, vp, vp, vp, i64, vp, ci, vp, vp, vp, vp, i64, i64, vp]
    ll = C.c_longlong
    L.ffhip_shard_range.argtypes = [ll, ci, ci, C.POINTER(ll), C.POINTER(ll)]
    L.ffhip_comm_unique_id.argtypes = [vp]
    L.ffhip_comm_init_rank.argtypes = [vp, ci, ci]
    L.ffhip_comm_init_rank.restype = vp
    L.ffhip_comm_destroy.argtypes = [vp]
    L.ffhip_comm_destroy.restype = None
    L.ffhip_batch_close.argtypes = [vp, ci, ci, ll, ll, ci, C.c_uint64, C.POINTER(BatchRecord), vp]
    L.ffhip_batch_complete.argtypes = [C.POINTER(BatchRecord), ci, ll]
    L.ffhip_vp8_filter_params.argtypes = [C.POINTER(Vp8FilterHeader), vp, C.POINTER(ci)]
    L.ffhip_bgra_checksum.argtypes = [vp, i64, i64, ci, ci, ci, vp, vp]
    L.ffhip_vp8_decode_frames.argtypes = [ci, ci, ci, vp, vp, vp, i64, vp, ci, vp, vp, ci, i64, vp, vp, vp, i64, i64, vp]
    L.ffhip_bgra_layout.argtypes = [C.POINTER(JpegGeom), C.POINTER(i64), C.POINTER(i64)]
    L.ffhip_env_value_test.argtypes = [C.c_char_p, vp, sz]
    L.ffhip_env_value_test.restype = C.c_long
    _lib = L
    return L


def reload_env():
    """The library reads its FFHIP_* switches once per process; after changing one in a live process, call this."""
    if _lib is not None:
        _lib.ffhip_reload_env()


def setenv(name, value):
    """Set (or, with None, remove) an FFHIP_* switch in this process and make the library read it again."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    reload_env()


def check(rc, what="ffhip call"):
    if rc != 0:
        raise FfhipError(f"{what} failed: {rc} ({lib().ffhip_strerror(rc).decode()})")


FFHIP_RETRIED = 1


def sync(stream=None):
    """ffhip_stream_sync: raises on an error code, returns the status otherwise -- 0, or FFHIP_RETRIED when the sync had to repeat a
    side-by-side VP8 call (outputs good; what the caller enqueued behind that call consumed the aborted run's and is stale)."""
    rc = lib().ffhip_stream_sync(stream)
    if rc < 0:
        check(rc, "ffhip_stream_sync")
    return rc


def require_device(device=0):
    """Bind to a gfx950 device or raise: the product path never degrades to the CPU."""
    L = lib()
    if L.ffhip_device_count() <= device:
        raise FfhipError("no HIP device visible: ffpic_amd needs an MI355X (gfx950); there is no CPU fallback")
    check(L.ffhip_init(device), f"ffhip_init({device})")
    return L
