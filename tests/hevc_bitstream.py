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


def pps(init_qp=30, transform_skip=1, transquant_bypass=1, sign_data_hiding=0, constrained_intra=0):
    w = BitWriter()
    w.ue(0); w.ue(0); w.u(1, 0); w.u(1, 0); w.u(3, 0); w.u(1, sign_data_hiding); w.u(1, 0); w.ue(0); w.ue(0); w.se(init_qp - 26)
    w.u(1, constrained_intra); w.u(1, transform_skip); w.u(1, 0)
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
    return rng.integers(1, 256, size=max(200000, n_bytes))[:n_bytes].astype(np.uint8).tobytes()   # (the first 200 000 values do not depend on the size drawn)


def stream(width, height, seed, n_bytes=8000, **pps_kw):
    return [vps(), sps(width, height), pps(**pps_kw), idr_slice(random_slice_data(seed, n_bytes))]


# ---- the same stream inside an ISO BMFF / HEIF container (a single-image .heic) -------------------------------------
# What the reference's loader reads (format/heif.c:466-540, format/basemedia.c): ftyp, then a meta box with hdlr, pitm,
# iloc, iinf / infe, iprp (ipco with hvcC + ispe, ipma); the coded image item sits in an mdat box as length-prefixed NAL
# units (decode_hvc1, heif.c:244-257), the parameter sets in the hvcC property (read_hvcc_box, heif.c:78-124).

def _box(fourcc, payload):
    return (8 + len(payload)).to_bytes(4, "big") + fourcc + payload


def _full_box(fourcc, version, flags, payload):
    return _box(fourcc, bytes([version]) + flags.to_bytes(3, "big") + payload)


def heic(width, height, seed, n_bytes=8000, **pps_kw):
    """bytes of a .heic file whose primary item is the hand-assembled intra picture stream(width, height, seed, ...)"""
    v, s, p, slice_nal = stream(width, height, seed, n_bytes, **pps_kw)
    # HEVCDecoderConfigurationRecord (ISO/IEC 14496-15 8.3.3.1): 23 bytes, then the parameter-set arrays
    cfg = (bytes([1, 0x01]) + (0x60000000).to_bytes(4, "big") + bytes([0x90, 0, 0, 0, 0, 0]) + bytes([120]) +
           bytes([0xF0, 0x00, 0xFC, 0xFD, 0xF8, 0xF8, 0x00, 0x00, 0x0F, 3]))
    assert len(cfg) == 23
    arrays = b"".join(bytes([0x80 | t]) + (1).to_bytes(2, "big") + len(n).to_bytes(2, "big") + n for t, n in ((32, v), (33, s), (34, p)))
    hvcc = _box(b"hvcC", cfg + arrays)
    ispe = _full_box(b"ispe", 0, 0, width.to_bytes(4, "big") + height.to_bytes(4, "big"))
    ipma = _full_box(b"ipma", 0, 0, (1).to_bytes(4, "big") + (1).to_bytes(2, "big") + bytes([2, 0x81, 0x02]))   # item 1: hvcC (essential), ispe
    iprp = _box(b"iprp", _box(b"ipco", hvcc + ispe) + ipma)
    hdlr = _full_box(b"hdlr", 0, 0, bytes(4) + b"pict" + bytes(12) + b"\0")
    pitm = _full_box(b"pitm", 0, 0, (1).to_bytes(2, "big"))
    iinf = _full_box(b"iinf", 0, 0, (1).to_bytes(2, "big") + _full_box(b"infe", 2, 0, (1).to_bytes(2, "big") + bytes(2) + b"hvc1" + b"\0"))
    payload = len(slice_nal).to_bytes(4, "big") + slice_nal
    ftyp = (24).to_bytes(4, "big") + b"ftyp" + b"heic" + bytes(4) + b"mif1" + b"heic"

    def meta_with(offset):
        iloc = _full_box(b"iloc", 0, 0, bytes([0x44, 0x00]) + (1).to_bytes(2, "big") +                      # offset / length 4 bytes, no base offset
                         (1).to_bytes(2, "big") + bytes(2) + (1).to_bytes(2, "big") + offset.to_bytes(4, "big") + len(payload).to_bytes(4, "big"))
        return _full_box(b"meta", 0, 0, hdlr + pitm + iloc + iinf + iprp)

    meta = meta_with(0)
    meta = meta_with(len(ftyp) + len(meta) + 8)     # the item's absolute file offset: behind ftyp, meta and the mdat header
    return ftyp + meta + _box(b"mdat", payload)


def heic_grid_1x1(width, height, seed, n_bytes=8000, **pps_kw):
    """bytes of a .heic file whose primary item is a 1 x 1 `grid` (decode_grid_items, heif.c:273-313) over ONE tile item:
    item 1 = the grid (8-byte ImageGrid payload in mdat, `dimg` reference to item 2), item 2 = the hvc1 tile.  With one
    tile the reference's "every tile into the same buffer" (heif.c:305) and a real compositor agree.
    Returns (file bytes, grid payload)."""
    v, s, p, slice_nal = stream(width, height, seed, n_bytes, **pps_kw)
    cfg = (bytes([1, 0x01]) + (0x60000000).to_bytes(4, "big") + bytes([0x90, 0, 0, 0, 0, 0]) + bytes([120]) +
           bytes([0xF0, 0x00, 0xFC, 0xFD, 0xF8, 0xF8, 0x00, 0x00, 0x0F, 3]))
    arrays = b"".join(bytes([0x80 | t]) + (1).to_bytes(2, "big") + len(n).to_bytes(2, "big") + n for t, n in ((32, v), (33, s), (34, p)))
    hvcc = _box(b"hvcC", cfg + arrays)
    ispe = _full_box(b"ispe", 0, 0, width.to_bytes(4, "big") + height.to_bytes(4, "big"))
    # properties: 1 = hvcC, 2 = ispe (tile), 3 = ispe (grid output, the same size); item 1 (grid): ispe 3; item 2 (tile): hvcC + ispe 2
    ipma = _full_box(b"ipma", 0, 0, (2).to_bytes(4, "big") + (1).to_bytes(2, "big") + bytes([1, 0x03]) + (2).to_bytes(2, "big") + bytes([2, 0x81, 0x02]))
    iprp = _box(b"iprp", _box(b"ipco", hvcc + ispe + ispe) + ipma)
    hdlr = _full_box(b"hdlr", 0, 0, bytes(4) + b"pict" + bytes(12) + b"\0")
    pitm = _full_box(b"pitm", 0, 0, (1).to_bytes(2, "big"))
    infe = lambda item_id, typ: _full_box(b"infe", 2, 0, item_id.to_bytes(2, "big") + bytes(2) + typ + b"\0")
    iinf = _full_box(b"iinf", 0, 0, (2).to_bytes(2, "big") + infe(1, b"grid") + infe(2, b"hvc1"))
    iref = _full_box(b"iref", 0, 0, _box(b"dimg", (1).to_bytes(2, "big") + (1).to_bytes(2, "big") + (2).to_bytes(2, "big")))
    grid = bytes([0, 0, 0, 0]) + width.to_bytes(2, "big") + height.to_bytes(2, "big")   # version, flags (16-bit sizes), rows - 1, columns - 1, output size
    tile = len(slice_nal).to_bytes(4, "big") + slice_nal
    ftyp = (24).to_bytes(4, "big") + b"ftyp" + b"heic" + bytes(4) + b"mif1" + b"heic"

    def meta_with(off):
        item = lambda item_id, o, n: item_id.to_bytes(2, "big") + bytes(2) + (1).to_bytes(2, "big") + o.to_bytes(4, "big") + n.to_bytes(4, "big")
        iloc = _full_box(b"iloc", 0, 0, bytes([0x44, 0x00]) + (2).to_bytes(2, "big") + item(1, off, len(grid)) + item(2, off + len(grid), len(tile)))
        return _full_box(b"meta", 0, 0, hdlr + pitm + iloc + iinf + iref + iprp)

    meta = meta_with(0)
    meta = meta_with(len(ftyp) + len(meta) + 8)
    return ftyp + meta + _box(b"mdat", grid + tile), grid
