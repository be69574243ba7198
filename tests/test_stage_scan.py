"""CPU: the host staging pass of the device entropy path (unstuffing + restart-marker detection in one pass, 16 bytes
at a time) against a byte-by-byte model, on streams dense in 0xFF / 0x00 / RSTn / other markers, with a guard band
behind the destination that must stay untouched.  Needs no GPU."""
import ctypes as C

import numpy as np
import pytest

from ffpic_amd import capi


def model(src, n_seg, raw=None):
    out, seg, i, k = bytearray(), [0], 0, 0
    def pad():
        ln = len(out) - seg[-1]
        if raw is not None:
            raw.append(ln)
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
    lib.ffhip_jpeg_stage_scan_raw_test.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t), C.c_void_p]
    lib.ffhip_jpeg_stage_scan_raw_test.restype = C.c_int
    lib.ffhip_jpeg_lut_test.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.ffhip_jpeg_lut_test.restype = C.c_int
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


def test_stage_scan_reports_the_intervals_own_lengths(L):
    """ffhip_jpeg_stage_scan_raw_test: raw[k] = bytes of interval k without its padding -- the subsequence decoder's lanes are cut from these"""
    rng = np.random.default_rng(77)
    for trial in range(400):
        n = int(rng.integers(0, 300))
        src = rng.integers(0, 256, size=n).astype(np.uint8)
        if n:
            src[rng.integers(0, n, size=max(1, n // 25))] = 0xFF
            for k in rng.integers(0, max(1, n - 1), size=3):
                if src[k] == 0xFF and k + 1 < n:
                    src[k + 1] = int(rng.choice([0x00, 0xD0, 0xD1, 0xD5]))
        src = np.ascontiguousarray(src)
        n_seg = int(rng.integers(1, 7))
        dst = np.zeros(n + 8 * n_seg + 128, np.uint8)
        seg = np.zeros(n_seg + 1, np.uint32)
        raw = np.full(n_seg + 1, 0xDEADBEEF, np.uint32)
        clean = C.c_size_t()
        found = L.ffhip_jpeg_stage_scan_raw_test(dst.ctypes.data, src.ctypes.data, n, seg.ctypes.data, n_seg, C.byref(clean), raw.ctypes.data)
        wraw = []
        want, wseg = model(bytes(src), n_seg, wraw)
        assert found == len(wseg) and list(raw[:found]) == wraw and raw[n_seg] == 0xDEADBEEF, trial
        ends = wseg[1:] + [len(want)]
        for k in range(found):                    # data, then zeros up to the next interval: at least four of them
            assert ends[k] - wseg[k] - wraw[k] >= 4 and not any(want[wseg[k] + wraw[k]:ends[k]])


def _dht_tables(data):
    """(class, id) -> (counts[16], values) of a JPEG file's DHT segments"""
    out, p = {}, 2
    while p + 4 <= len(data):
        m, ln = data[p + 1], (data[p + 2] << 8) | data[p + 3]
        if m == 0xDA:
            break
        if m == 0xC4:
            s, i = data[p + 4:p + 2 + ln], 0
            while i + 17 <= len(s):
                counts = list(s[i + 1:i + 17])
                n = sum(counts)
                out[(s[i] >> 4, s[i] & 15)] = (counts, list(s[i + 17:i + 17 + n]))
                i += 17 + n
        p += 2 + ln
    return out


@pytest.mark.parametrize("name", ["file_q85_420.jpg", "file_q92_444.jpg", "file_q80_grey.jpg"])
def test_device_lookup_table_decodes_every_code_and_nothing_else(L, name):
    """the two-level table of ffhip_huff_gpu.hip (build_lut), built on the host: every canonical code of the file's tables, followed by any bits, looks up
    to its own (length, symbol) -- short codes in level one, long ones through their prefix's group --, and 16 bits that start no code look up to the
    "no such code" entry, never to a symbol (the subsequence decoder reads from wrong bit positions by design and relies on that)"""
    import os
    data = open(os.path.join(os.path.dirname(__file__), "golden", name), "rb").read()
    buf = np.frombuffer(data, dtype=np.uint8)
    for (tc, th), (counts, vals) in _dht_tables(data).items():
        lut = np.zeros(1536, np.uint16)
        assert L.ffhip_jpeg_lut_test(buf.ctypes.data, buf.size, tc * 4 + th, lut.ctypes.data) == 0

        def look(bits16):
            e = int(lut[bits16 >> 7])
            if e & 0x8000:
                e = int(lut[512 + ((e & 0xff) << 7) + (bits16 & 127)])
            return e
        code, k, is_code = 0, 0, np.zeros(1 << 16, bool)
        for ln in range(1, 17):
            for _ in range(counts[ln - 1]):
                first = code << (16 - ln)
                for tail in (0, (1 << (16 - ln)) - 1, (0x5a5a >> ln) & ((1 << (16 - ln)) - 1)):
                    e = look(first | tail)
                    assert (e >> 8) & 31 == ln and e & 0xff == vals[k] and not e & 0x4000, (name, tc, th, ln, hex(first | tail))
                is_code[first:first + (1 << (16 - ln))] = True
                code += 1
                k += 1
            code <<= 1
        rest = np.flatnonzero(~is_code)
        for bits16 in rest[:: max(1, len(rest) // 500)]:
            assert look(int(bits16)) == 0x5000, (name, tc, th, hex(int(bits16)))
