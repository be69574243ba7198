"""ctypes bindings to the CPU checker (oracle/libffo.so) and, where it was built,
the reference itself (oracle/_ref/libffpic_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (ffpic_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
FFO_SO = os.path.join(ORACLE_DIR, "libffo.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libffpic_ref.so")

i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class Geom(C.Structure):
    """Same layout as ffo_jpeg_geom / ffhip_jpeg_geom."""
    _fields_ = [("mcu_cols", C.c_int32), ("mcu_rows", C.c_int32), ("ncomp", C.c_int32),
                ("h", C.c_int32), ("v", C.c_int32), ("qt_id", C.c_int32 * 3)]

    def as_array(self):
        return np.array([self.mcu_cols, self.mcu_rows, self.ncomp, self.h, self.v,
                         self.qt_id[0], self.qt_id[1], self.qt_id[2]], dtype=np.int32)

    @property
    def width(self):
        return self.mcu_cols * 8 * self.h

    @property
    def height(self):
        return self.mcu_rows * 8 * self.v

    @property
    def mcus(self):
        return self.mcu_cols * self.mcu_rows


def make_geom(mcu_cols, mcu_rows, ncomp=3, h=2, v=2, qt_id=(0, 1, 1)):
    g = Geom()
    g.mcu_cols, g.mcu_rows, g.ncomp, g.h, g.v = mcu_cols, mcu_rows, ncomp, h, v
    for i in range(3):
        g.qt_id[i] = qt_id[i]
    return g


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


_ffo = None
_ref = None


def ffo():
    """Load (building on demand) our CPU restatement."""
    global _ffo
    if _ffo is None:
        if not os.path.exists(FFO_SO):
            build_oracle()
        L = C.CDLL(FFO_SO)
        L.ffo_jpeg_dequant.argtypes = [i16p, i16p, u16p, C.c_int]
        L.ffo_idct_8x8_16.argtypes = [i16p]
        L.ffo_yuv_to_bgra32_mcu16.argtypes = [u8p, C.c_int, i16p, i16p, i16p, C.c_int, C.c_int]
        L.ffo_jpeg_recon_image.argtypes = [C.POINTER(Geom), i16p, C.c_void_p, C.c_void_p, u16p, u8p, C.c_int64]
        L.ffo_jpeg_recon_image.restype = C.c_int
        L.ffo_jpeg_recon_batch.argtypes = [C.POINTER(Geom), C.c_int, i16p, C.c_void_p, C.c_void_p, u16p,
                                           C.c_int64, u8p, C.c_int64, C.c_int64, C.c_int]
        L.ffo_jpeg_recon_batch.restype = C.c_int
        L.ffo_vp8_idct_4x4.argtypes = [i16p]
        L.ffo_vp8_iwht_long.argtypes = [i16p, i16p]
        L.ffo_vp8_iwht_fast.argtypes = [i16p, i16p]
        L.ffo_vp8_residual_mb.argtypes = [i16p, u8p, C.c_int, u16p, i16p]
        L.ffo_vp8_recon_frame.argtypes = [C.c_int, C.c_int, u8p, i16p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ffo_vp8_loopfilter_frame.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p]
        L.ffo_hevc_intra_tu.argtypes = [C.c_void_p, i16p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ffo_hevc_intra_recon.argtypes = [C.c_void_p, C.c_long, i16p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_hevc_idct_4x4_dst.argtypes = [i16p, i16p, C.c_int, C.c_int]
        L.ffo_hevc_scale.argtypes = [i16p, i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ffo_hevc_transform.argtypes = [i16p, i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_hevc_residual_tu.argtypes = [i16p, i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ffo_yuv420_to_bgra32.argtypes = [u8p, C.c_int, u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ffo_yuv420_to_bgra32_16bit.argtypes = [u8p, C.c_int, i16p, i16p, i16p, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int]
        L.ffo_yuv400_to_bgra32_16bit.argtypes = [u8p, C.c_int, i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        _ffo = L
    return _ffo


def have_ref():
    return os.path.exists(REF_SO) or os.path.isdir("/root/reference")


def ref():
    """Load the compiled reference (only exists if it was built in the build container)."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            build_ref()
        L = C.CDLL(REF_SO)
        L.ref_jpeg_dequant.argtypes = [i16p, i16p, u16p, C.c_int]
        L.ref_idct_8x8_16.argtypes = [i16p]
        L.ref_yuv_to_bgra32_mcu16.argtypes = [u8p, C.c_int, i16p, i16p, i16p, C.c_int, C.c_int]
        L.ref_jpeg_recon_image.argtypes = [i32p, i16p, C.c_void_p, C.c_void_p, u16p, u8p, C.c_int64]
        L.ref_vp8_idct_4x4.argtypes = [i16p]
        L.ref_vp8_iwht_long.argtypes = [i16p, i16p]
        L.ref_vp8_iwht_fast.argtypes = [i16p, i16p]
        L.ref_vp8_residual_mb.argtypes = [i16p, u8p, C.c_int, u16p, i16p]
        L.ref_vp8_residual_blocks_driven.argtypes = [u8p, C.c_int, C.c_int, u8p, u8p, u16p, u8p, i16p, u8p, i16p]
        L.ref_webp_filter_params.argtypes = [i32p, i32p]
        L.ref_vp8_recon_frame.argtypes = [C.c_int, C.c_int, u8p, i16p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_vp8_loopfilter_frame.argtypes = [C.c_int, C.c_int, C.c_int, u8p, u8p, u8p, u8p, u8p]
        L.ref_hevc_intra_tu.argtypes = [C.c_int] * 6 + [C.c_uint64, C.c_uint64, i16p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.idct_4x4_hevc.argtypes = [i16p, i16p, C.c_int, C.c_bool]
        L.ref_hevc_scale.argtypes = [i16p, i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ref_hevc_transform.argtypes = [i16p, i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_hevc_scale_and_transform.argtypes = [i16p, i16p] + [C.c_int] * 9 + [C.c_void_p]
        L.ref_hevc_scale_and_transform.restype = C.c_int
        L.YUV420_to_BGRA32.argtypes = [u8p, C.c_int, u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.YUV420_to_BGRA32_16bit.argtypes = [u8p, C.c_int, i16p, i16p, i16p, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int]
        L.YUV400_to_BGRA32_16bit.argtypes = [u8p, C.c_int, i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        _ref = L
    return _ref


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- JPEG helpers

def oracle_jpeg_recon(g, coef_y, coef_u, coef_v, quant, n_images=1, n_threads=1):
    """Run the CPU restatement over a batch; returns BGRA [n, H, W, 4]."""
    H, W = g.height, g.width
    out = np.zeros((n_images, H, W, 4), dtype=np.uint8)
    quant = np.ascontiguousarray(quant, dtype=np.uint16)
    qstride = 0 if quant.ndim == 2 else 4 * 64
    rc = ffo().ffo_jpeg_recon_batch(C.byref(g), n_images, coef_y, _vp(coef_u), _vp(coef_v), quant,
                                    qstride, out.reshape(-1), W * 4, H * W * 4, n_threads)
    assert rc == 0, rc
    return out


def ref_jpeg_recon(g, coef_y, coef_u, coef_v, quant):
    """Run the compiled reference over ONE image; returns BGRA [H, W, 4]."""
    H, W = g.height, g.width
    out = np.zeros((H, W, 4), dtype=np.uint8)
    quant = np.ascontiguousarray(quant, dtype=np.uint16)
    ref().ref_jpeg_recon_image(g.as_array(), coef_y, _vp(coef_u), _vp(coef_v), quant, out.reshape(-1), W * 4)
    return out


# ---------------------------------------------------------------- VP8 frame helpers

def _vp8_frame(fn, mbcols, mbrows, modes, residual, resmap, fill=0):
    """Run a whole-frame predict+recon; planes get one zeroed guard row in front (what the
    reference's 16x16 V/H predictors may read at the top row / left column)."""
    ys, uvs = 16 * mbcols, 8 * mbcols
    yb = np.full((16 * mbrows + 1) * ys, fill, np.uint8)
    ub = np.full((8 * mbrows + 1) * uvs, fill, np.uint8)
    vb = np.full((8 * mbrows + 1) * uvs, fill, np.uint8)
    yb[:ys] = 0; ub[:uvs] = 0; vb[:uvs] = 0
    rm = None if resmap is None else np.ascontiguousarray(resmap, dtype=np.int32).ctypes.data_as(C.c_void_p)
    fn(mbcols, mbrows, np.ascontiguousarray(modes), np.ascontiguousarray(residual).reshape(-1), rm,
       C.c_void_p(yb.ctypes.data + ys), C.c_void_p(ub.ctypes.data + uvs), C.c_void_p(vb.ctypes.data + uvs))
    return (yb[ys:].reshape(16 * mbrows, ys), ub[uvs:].reshape(8 * mbrows, uvs), vb[uvs:].reshape(8 * mbrows, uvs))


def oracle_vp8_frame(mbcols, mbrows, modes, residual, resmap=None):
    return _vp8_frame(ffo().ffo_vp8_recon_frame, mbcols, mbrows, modes, residual, resmap)


def ref_vp8_frame(mbcols, mbrows, modes, residual, resmap=None):
    return _vp8_frame(ref().ref_vp8_recon_frame, mbcols, mbrows, modes, residual, resmap)


def ref_vp8_driven(n_mb, seed, regime="random"):
    """The reference's vp8_decode_residual_block (format/webp.c:1125-1199) run over n_mb macroblocks of one row from a
    synthetic bool-decoder state (seeded random bytes and coefficient probabilities; oracle/ref_statics_webp.c::
    ref_vp8_residual_blocks_driven).  Returns what ffhip_vp8_residual_batch takes -- levels [n][25][16], info [n][32],
    quant [4][8] -- and the residual [n][384] the reference wrote."""
    from ffpic_amd import synth
    rng = np.random.default_rng(0x7B8 + seed)
    by = rng.integers(0, 256, size=n_mb * 4096, dtype=np.uint8)
    by[0] &= 0x7F                      # a first byte of 255 leaves this bool decoder with value == range for good
    modes = np.where(rng.random(n_mb) < 0.4, 4, rng.integers(0, 4, size=n_mb)).astype(np.uint8)
    seg = rng.integers(0, 4, size=n_mb).astype(np.uint8)
    q = synth.vp8_quant(seed=seed)
    probs = rng.integers(1, 256, size=(4, 8, 3, 11)).astype(np.uint8)
    if regime == "sparse":             # node 0 decides "end of block": likely
        probs[..., 0] = rng.integers(160, 256, size=(4, 8, 3))
    elif regime == "dense":            # rarely end of block, rarely zero: long blocks with large tokens
        probs[..., 0] = rng.integers(1, 40, size=(4, 8, 3))
        probs[..., 1] = rng.integers(1, 80, size=(4, 8, 3))
    lv = np.zeros((n_mb, 25, 16), np.int16)
    nz = np.zeros((n_mb, 25), np.uint8)
    dst = np.zeros((n_mb, 384), np.int16)
    rc = ref().ref_vp8_residual_blocks_driven(by, by.size, n_mb, modes, seg, q, probs, lv, nz, dst)
    assert rc == 0, rc
    info = np.zeros((n_mb, 32), np.uint8)
    info[:, :25] = nz
    info[:, 25] = modes != 4
    info[:, 26] = seg
    return lv, info, q, dst


# ---------------------------------------------------------------- HEVC intra helpers

def _hevc_planes(width, height, chroma, fill=0, csub=2):
    py = np.full((height, width), fill, np.int16)
    if chroma:
        return (py, np.full((height // csub, width // csub), fill, np.int16),
                np.full((height // csub, width // csub), fill, np.int16))
    return py, np.zeros((1, 1), np.int16), np.zeros((1, 1), np.int16)


def oracle_hevc_intra(tus, residual, width, height, chroma=True, bd_y=8, bd_c=8, csub=2):
    py, pu, pv = _hevc_planes(width, height, chroma, csub=csub)
    tus = np.ascontiguousarray(tus)
    ffo().ffo_hevc_intra_recon(tus.ctypes.data_as(C.c_void_p), len(tus), np.ascontiguousarray(residual),
                               py.ctypes.data_as(C.c_void_p), pu.ctypes.data_as(C.c_void_p),
                               pv.ctypes.data_as(C.c_void_p), width, max(width // csub, 1), bd_y, bd_c)
    return py, pu, pv


def ref_hevc_intra(tus, residual, width, height, chroma=True, bd_y=8, bd_c=8, csub=2):
    py, pu, pv = _hevc_planes(width, height, chroma, csub=csub)
    R = ref()
    planes = (py, pu, pv)
    residual = np.ascontiguousarray(residual)
    for t in tus:
        pl = planes[int(t["cidx"])]
        n = 1 << int(t["log2_size"])
        ro = int(t["res_offset"])
        blk = np.ascontiguousarray(residual[ro:ro + n * n]) if int(t["flags"]) & 2 else np.zeros(n * n, np.int16)
        R.ref_hevc_intra_tu(int(t["x"]), int(t["y"]), int(t["log2_size"]), int(t["cidx"]), int(t["pred_mode"]),
                            int(t["flags"]), int(t["avail_top"]), int(t["avail_left"]), blk,
                            pl.ctypes.data_as(C.c_void_p), pl.shape[1], bd_y, bd_c, int(t["res_scale"]))
    return py, pu, pv
