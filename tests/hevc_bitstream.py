"""Hand-assembled HEVC intra bitstreams for fixture generation (tests/golden/make_golden.py::gen_hevc_file).

TEST INFRASTRUCTURE.  There is no HEIF/HEVC encoder in the image, so the parameter sets and the slice
header are written bit by bit from ITU-T H.265 7.3 (what the reference's parse_vps / parse_sps / parse_pps /
parse_slice_segment_header read, coding/hevc.c:744-1170, 2252-2330, 2660-3190), and the slice DATA is seeded random
bytes: any byte string is a decodable CABAC stream, so the reference's own parser turns it into a random but perfectly
valid intra picture -- coding-quadtree splits, all 35 modes, 4x4 .. 32x32 transform units, transform skip, transquant
bypass, large coefficients.  What makes a stream usable is only that the reference's slice loop finds
end_of_slice_segment_flag == 1 right after the last coding tree unit (hevc.c:7007-7019); seeds for which it does were
found by search and are listed by the generator.
"""
import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def ue(self, v):
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def trailing(self):
        self.bits.append(1)
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return bytes(int("".join(map(str, self.bits[i:i + 8])), 2) for i in range(0, len(self.bits), 8))


def escape(rbsp):
    """emulation prevention (7.4.2): 00 00 0x -> 00 00 03 0x for x <= 3"""
    out, zeros = bytearray(), 0
    for b in rbsp:
        if zeros >= 2 and b <= 3:
            out.append(3)
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def nal(nal_unit_type, rbsp):
    return bytes([nal_unit_type << 1, 1]) + escape(rbsp)      # nuh_layer_id 0, nuh_temporal_id_plus1 1


def _profile_tier_level(w):
    w.u(2, 0); w.u(1, 0); w.u(5, 1); w.u(32, 0x60000000)      # Main, compatible with profiles 1 and 2
    w.u(4, 0b1001); w.u(32, 0); w.u(12, 0); w.u(8, 120)       # progressive + frame-only, 44 reserved bits, level 4


def vps():
    w = BitWriter()
    w.u(4, 0); w.u(1, 1); w.u(1, 1); w.u(6, 0); w.u(3, 0); w.u(1, 1); w.u(16, 0xFFFF)
    _profile_tier_level(w)
    w.u(1, 1); w.ue(0); w.ue(0); w.ue(0); w.u(6, 0); w.ue(0); w.u(1, 0); w.u(1, 0)
    w.trailing()
    return nal(32, w.bytes())


def sps(width, height, bitdepth=8, ctb_log2=6, min_cb_log2=3, min_tb_log2=2, max_tb_log2=5, depth_intra=3, strong=1):
    w = BitWriter()
    w.u(4, 0); w.u(3, 0); w.u(1, 1)
    _profile_tier_level(w)
    w.ue(0); w.ue(1); w.ue(width); w.ue(height); w.u(1, 0); w.ue(bitdepth - 8); w.ue(bitdepth - 8); w.ue(4)
    w.u(1, 1); w.ue(0); w.ue(0); w.ue(0)
    w.ue(min_cb_log2 - 3); w.ue(ctb_log2 - min_cb_log2); w.ue(min_tb_log2 - 2); w.ue(max_tb_log2 - min_tb_log2); w.ue(0); w.ue(depth_intra)
    w.u(1, 0)                                  # scaling_list_enabled_flag
    w.u(1, 0); w.u(1, 0); w.u(1, 0)            # amp, sample adaptive offset, pcm
    w.ue(0); w.u(1, 0); w.u(1, 0); w.u(1, strong); w.u(1, 0); w.u(1, 0)
    w.trailing()
    return nal(33, w.bytes())


def pps(init_qp=30, transform_skip=1, transquant_bypass=1, sign_data_hiding=0):
    w = BitWriter()
    w.ue(0); w.ue(0); w.u(1, 0); w.u(1, 0); w.u(3, 0); w.u(1, sign_data_hiding); w.u(1, 0); w.ue(0); w.ue(0); w.se(init_qp - 26)
    w.u(1, 0); w.u(1, transform_skip); w.u(1, 0)
    w.se(0); w.se(0); w.u(1, 0); w.u(1, 0); w.u(1, 0); w.u(1, transquant_bypass); w.u(1, 0); w.u(1, 0)
    w.u(1, 0); w.u(1, 0); w.u(1, 0); w.u(1, 0); w.ue(0); w.u(1, 0); w.u(1, 0)
    w.trailing()
    return nal(34, w.bytes())


def idr_slice(slice_data):
    w = BitWriter()
    w.u(1, 1); w.u(1, 0); w.ue(0); w.ue(2); w.se(0)       # first slice, pps 0, I slice, slice_qp_delta 0
    w.trailing()                                           # byte_alignment()
    return nal(19, w.bytes() + slice_data)                 # IDR_W_RADL


def random_slice_data(seed, n_bytes):
    """no zero bytes: nothing to escape, and the stream stays the same under the reference's 00 00 03 removal"""
    rng = np.random.default_rng(seed)
    return bytes(int(x) for x in rng.integers(1, 256, size=200000)[:n_bytes])


def stream(width, height, seed, n_bytes=8000, **pps_kw):
    return [vps(), sps(width, height), pps(**pps_kw), idr_slice(random_slice_data(seed, n_bytes))]
