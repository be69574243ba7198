"""CPU: the host staging pass of the device entropy path (unstuffing + restart-marker detection in one pass, 16 bytes
at a time) against a byte-by-byte model, on streams dense in 0xFF / 0x00 / RSTn / other markers, with a guard band
behind the destination that must stay untouched.  Needs no GPU."""
import ctypes as C

import numpy as np
import pytest

from ffpic_amd import capi


def model(src, n_seg):
    out, seg, i, k = bytearray(), [0], 0, 0
    def pad():
        ln = len(out) - seg[-1]
        out.extend(b"\0" * ((((ln + 3) & ~3) + 4) - ln))
    while True:
        while i < len(src) and src[i] != 0xFF:
            out.append(src[i]); i += 1
        if i + 1 >= len(src):
            break
        b = src[i + 1]
        if b == 0:
            out.append(0xFF); i += 2; continue
        if not (0xD0 <= b <= 0xD7) or k + 1 >= n_seg:
            break
        pad(); k += 1; seg.append(len(out)); i += 2
    pad()
    return bytes(out), seg


@pytest.fixture(scope="module")
def L():
    lib = capi.lib()
    lib.ffhip_jpeg_stage_scan_test.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t)]
    lib.ffhip_jpeg_stage_scan_test.restype = C.c_int
    return lib


@pytest.mark.parametrize("seed", range(12))
def test_stage_scan_matches_model(L, seed):
    rng = np.random.default_rng(seed)
    for trial in range(300):
        n = int(rng.integers(0, 400))
        alphabet = np.array([0xFF, 0xFF, 0xFF, 0x00, 0x00, 0xD0, 0xD3, 0xD7, 0xD9, 0xC4, 0x12, 0x80, 0xFE], np.uint8)
        dense = rng.random() < 0.5
        src = rng.choice(alphabet, size=n) if dense else rng.integers(0, 256, size=n).astype(np.uint8)
        if not dense and n:
            src[rng.integers(0, n, size=max(1, n // 40))] = 0xFF
        src = np.ascontiguousarray(src)
        n_seg = int(rng.integers(1, 9))
        cap = n + 8 * n_seg + 64
        dst = np.full(cap + 64, 0xA5, np.uint8)
        seg = np.full(n_seg + 4, 0xDEADBEEF, np.uint32)
        clean = C.c_size_t()
        found = L.ffhip_jpeg_stage_scan_test(dst.ctypes.data, src.ctypes.data, n, seg.ctypes.data, n_seg, C.byref(clean))
        want, wseg = model(bytes(src), n_seg)
        assert found == len(wseg), (seed, trial)
        assert clean.value == len(want) and bytes(dst[:clean.value]) == want, (seed, trial)
        assert list(seg[:found]) == wseg and (seg[n_seg:] == 0xDEADBEEF).all()
        assert (dst[cap:] == 0xA5).all(), "wrote past the reserved slack"
