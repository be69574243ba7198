"""Minimal baseline-JPEG entropy decoder (Huffman, interleaved scan, restart markers)
used ONLY to build and check test fixtures: it turns a .jpg into what the
reconstruction stage consumes -- quantised coefficients, natural order, int16,
blocks in MCU order per component -- plus natural-order quant tables, i.e. the
data format/jpg.c holds after decode_data_unit (jpg.c:521-539) and read_dqt
(jpg.c:78-105).  Written from ITU-T T.81; test infrastructure, not product code.
"""
import numpy as np


def _zigzag_order():
    order = sorted(range(64), key=lambda i: (i // 8 + i % 8, (i // 8) if (i // 8 + i % 8) % 2 else (i % 8)))
    return np.array(order)


ZZ = _zigzag_order()  # ZZ[k] = natural index of the k-th coefficient in scan order


class _Bits:
    def __init__(self, data):
        self.d, self.p, self.acc, self.n = data, 0, 0, 0

    def _fill(self):
        b = self.d[self.p]
        self.p += 1
        if b == 0xFF:
            nxt = self.d[self.p]
            if nxt == 0:
                self.p += 1          # stuffed zero
            else:
                self.p -= 1          # a marker: feed zeros, do not advance
                b = 0
        self.acc = (self.acc << 8) | b
        self.n += 8

    def bit(self):
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def reset(self):
        """byte-align and step over an RSTn marker"""
        self.acc = self.n = 0
        while not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _build_huff(counts, symbols):
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[(length, code)] = symbols[k]
            code += 1
            k += 1
        code <<= 1
    return table


def _decode_sym(br, table):
    code = 0
    for length in range(1, 17):
        code = (code << 1) | br.bit()
        s = table.get((length, code))
        if s is not None:
            return s
    raise ValueError("bad Huffman code")


def _extend(v, t):
    return v - (1 << t) + 1 if t and v < (1 << (t - 1)) else v


def decode(data):
    """-> dict(mcu_cols, mcu_rows, ncomp, h, v, qt_id, quant[4][64] uint16, coef[3] int16 flat or None)"""
    data = bytes(data)
    assert data[:2] == b"\xff\xd8"
    p = 2
    quant = np.ones((4, 64), dtype=np.uint16)
    huff = {}
    comps, restart = [], 0
    while True:
        assert data[p] == 0xFF
        m = data[p + 1]
        p += 2
        if m == 0xD9:
            raise ValueError("EOI before SOS")
        L = (data[p] << 8) | data[p + 1]
        seg = data[p + 2:p + L]
        p += L
        if m == 0xDB:
            i = 0
            while i < len(seg):
                prec, tid = seg[i] >> 4, seg[i] & 15
                i += 1
                for k in range(64):
                    if prec:
                        quant[tid][ZZ[k]] = (seg[i] << 8) | seg[i + 1]
                        i += 2
                    else:
                        quant[tid][ZZ[k]] = seg[i]
                        i += 1
        elif m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1:i + 17])
                n = sum(counts)
                huff[(tc, th)] = _build_huff(counts, list(seg[i + 17:i + 17 + n]))
                i += 17 + n
        elif m in (0xC0, 0xC1):
            height, width = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            for c in range(seg[5]):
                cid, hv, tq = seg[6 + 3 * c:9 + 3 * c]
                comps.append(dict(id=cid, h=hv >> 4, v=hv & 15, tq=tq))
        elif m == 0xC2:
            raise ValueError("progressive JPEG not supported by this fixture helper")
        elif m == 0xDD:
            restart = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            ns = seg[0]
            assert ns == len(comps), "non-interleaved scans not supported"
            for c in range(ns):
                cs, t = seg[1 + 2 * c], seg[2 + 2 * c]
                comp = next(x for x in comps if x["id"] == cs)
                comp["td"], comp["ta"] = t >> 4, t & 15
            break
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    mcu_cols = -(-width // (8 * hmax))
    mcu_rows = -(-height // (8 * vmax))
    for c in comps[1:]:
        assert c["h"] == 1 and c["v"] == 1, "chroma must be 1x1 (reference colour converter)"
    br = _Bits(data[p:] + b"\xff\xd9\x00\x00")
    planes = [np.zeros((mcu_cols * mcu_rows * c["h"] * c["v"], 64), dtype=np.int16) for c in comps]
    pred = [0] * len(comps)
    count = 0
    for mcu in range(mcu_cols * mcu_rows):
        if restart and count == restart:
            br.reset()
            pred = [0] * len(comps)
            count = 0
        count += 1
        for ci, c in enumerate(comps):
            dc_t, ac_t = huff[(0, c["td"])], huff[(1, c["ta"])]
            for b in range(c["h"] * c["v"]):
                blk = planes[ci][mcu * c["h"] * c["v"] + b]
                t = _decode_sym(br, dc_t)
                pred[ci] += _extend(br.bits(t), t)
                blk[0] = pred[ci]
                k = 1
                while k < 64:
                    rs = _decode_sym(br, ac_t)
                    r, s = rs >> 4, rs & 15
                    if s == 0:
                        if r == 15:
                            k += 16
                            continue
                        break
                    k += r
                    blk[ZZ[k]] = _extend(br.bits(s), s)
                    k += 1
    coef = [np.ascontiguousarray(pl.reshape(-1)) for pl in planes] + [None] * (3 - len(comps))
    qt = [c["tq"] for c in comps] + [0] * (3 - len(comps))
    return dict(mcu_cols=mcu_cols, mcu_rows=mcu_rows, ncomp=len(comps), h=comps[0]["h"], v=comps[0]["v"],
                qt_id=tuple(qt), quant=quant, coef=coef, width=width, height=height)
