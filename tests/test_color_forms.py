"""CPU: the arithmetic identities the fused kernel relies on, proven by enumeration.

* float-assisted exact floor division used for the chroma terms
  (ffhip_jpeg.hip::fdiv_f32): (int)((float)(2x+1) * fl(1/(2d))) == x // d on the whole
  range each divisor is used with (IEEE float32 multiply, emulated by numpy);
* the integer colour forms against the double arithmetic of utils/colorspace.c:162-164 on
  the IDCT output domain (tests/tools/check_color_int.c, subsampled here; the full sweep
  takes ~10 s on 8 cores and was run when the kernel was written);
* the IDCT output domain itself: (v >> 18) of any int32 lies in [-8192, 8191].
"""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fdiv_f32(two_x_plus_1, inv_2d):
    return (two_x_plus_1.astype(np.float32) * np.float32(inv_2d)).astype(np.int64)


def test_fdiv_ranges_exact():
    uu = np.arange(-128, 8064, dtype=np.int64)
    # fr = floor(32 vv / 25): numerator 64 vv + 2*25*164 + 1
    assert np.array_equal(fdiv_f32(64 * uu + 8201, np.float32(1.0) / np.float32(50.0)) - 164, (32 * uu) // 25)
    # fb = floor(266 uu / 125)
    assert np.array_equal(fdiv_f32(532 * uu + 68251, np.float32(1.0) / np.float32(250.0)) - 273, (266 * uu) // 125)
    # fg = floor(-(215 uu + 381 vv) / 1000) over every reachable numerator t = 4806000 - s
    t = np.arange(0, 4806000 + 596 * 128 + 1, dtype=np.int64)
    assert (2 * t + 1).max() < 2 ** 24
    assert np.array_equal(fdiv_f32(2 * t + 1, np.float32(1.0) / np.float32(2000.0)), t // 1000)
    smin, smax = 215 * -128 + 381 * -128, 215 * 8063 + 381 * 8063
    assert 4806000 - smax >= 0 and 4806000 - smin <= t.max()


def test_idct_output_domain():
    v = np.array([-2 ** 31, -1, 0, 2 ** 31 - 1], dtype=np.int64)
    assert ((v >> 18).min(), (v >> 18).max()) == (-8192, 8191)
    # packed 16-bit colour sums cannot overflow int16
    assert 8191 + (32 * 8063) // 25 < 32768 and 8191 + (266 * 8063) // 125 < 32768
    assert -(596 * 8063) // 1000 > -32768


def test_integer_colour_forms_subsampled(tmp_path):
    exe = tmp_path / "check_color_int"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "tools", "check_color_int.c"), "-o", str(exe)])
    out = subprocess.run([str(exe), "quick"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "R mismatches 0, B mismatches 0" in out.stdout
    assert "G mismatches on non-sensitive chroma: 0" in out.stdout


def test_fma_term_forms_exhaustive(tmp_path):
    """the float forms of the chroma terms the JPEG kernels compute (ffhip_jpeg.hip::chroma_term_bits: one fma and one add
    per term, the int16 result read from the float's low bits) against the integer definitions, for EVERY pair of raw
    chroma samples in [0, 8191]^2 (67 M pairs, under a second): tests/tools/check_color_fma.c"""
    exe = tmp_path / "check_color_fma"
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tests", "tools", "check_color_fma.c"), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fr mismatches 0, fb mismatches 0" in out.stdout and "fg mismatches 0, sensitivity mismatches 0" in out.stdout
