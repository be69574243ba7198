"""CPU: the synchronisation scheme of the device Huffman decoder (ffhip_huff_gpu.hip, "The subsequence decoder" in DESIGN.md 5) as a model in plain
Python, on the fixture files: a baseline scan cut into subsequences of a fixed number of bits, every subsequence decoded from a GUESSED state in round 0,
then from the exit the subsequence in front recorded, until nothing changes.  Claims checked here, without a GPU: the fixed point is the sequential decode
(bit position, block slot of the MCU and coefficient index at every boundary; blocks and DC sums per subsequence), it is reached in a handful of rounds at
the shipped length (4 for the 640x480 fixture's 238 subsequences of 2048 bits, 13 with 950 of 512 bits), and in no more rounds than there are
subsequences at any length.  The decode step follows format/jpg.c:255-415 (decode_data_unit) and
coding/huffman.c:92-222; the kernels' parity with the reference's planes is tests/test_huff_gpu.py's business."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def parse(data):
    """tables, sampling and the unstuffed scan bits of a baseline file without restart markers"""
    dht, p, comps, sos = {}, 2, [], None
    while p + 4 <= len(data):
        m, ln = data[p + 1], (data[p + 2] << 8) | data[p + 3]
        seg = data[p + 4:p + 2 + ln]
        if m == 0xC4:
            i = 0
            while i + 17 <= len(seg):
                counts = list(seg[i + 1:i + 17])
                n = sum(counts)
                code, k, lut = 0, 0, {}
                for length in range(1, 17):
                    for _ in range(counts[length - 1]):
                        lut[(length, code)] = seg[i + 17 + k]
                        code += 1
                        k += 1
                    code <<= 1
                dht[(seg[i] >> 4, seg[i] & 15)] = lut
                i += 17 + n
        elif m == 0xC0:
            comps = [(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15) for c in range(seg[5])]
            height, width = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
        elif m == 0xDD:
            pytest.skip("a file with restart markers: its intervals are separate scans")
        elif m == 0xDA:
            sos = {seg[1 + 2 * k]: (seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15) for k in range(seg[0])}
            p += 2 + ln
            break
        p += 2 + ln
    raw = bytearray()
    while p < len(data):
        if data[p] == 0xFF:
            if data[p + 1] == 0:
                raw.append(0xFF)
                p += 2
                continue
            break
        raw.append(data[p])
        p += 1
    bits = np.unpackbits(np.frombuffer(bytes(raw), dtype=np.uint8)).astype(np.uint8)
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcus = -(-width // (8 * hmax)) * -(-height // (8 * vmax))
    slots = []                                  # the MCU's blocks in scan order: (component, DC table, AC table)
    for ci, (cid, h, v) in enumerate(comps):
        nb = h * v if len(comps) > 1 else 1
        slots += [(ci, dht[(0, sos[cid][0])], dht[(1, sos[cid][1])])] * nb
    return bits, slots, mcus, len(comps)


def symbol(bits, pos, lut):
    """one Huffman symbol at bit `pos`: (symbol, length); bits behind the end read as zero; no code of any length: (0, 16), as the kernels' LUT_NO_CODE"""
    code = 0
    for length in range(1, 17):
        code = (code << 1) | (int(bits[pos + length - 1]) if pos + length - 1 < len(bits) else 0)
        if (length, code) in lut:
            return lut[(length, code)], length
    return 0, 16


def take(bits, pos, n):
    v = 0
    for i in range(n):
        v = (v << 1) | (int(bits[pos + i]) if pos + i < len(bits) else 0)
    return v


def run(bits, slots, state, limit):
    """decode from state = (pos, slot, k) until pos >= limit: the exit state, blocks completed, DC differences summed per component"""
    pos, slot, k = state
    blocks, dcs = 0, [0, 0, 0]
    while pos < limit:
        comp, dc_lut, ac_lut = slots[slot]
        if k == 0:
            s, ln = symbol(bits, pos, dc_lut)
            s &= 15
            v = take(bits, pos + ln, s)
            if s and v < (1 << (s - 1)):
                v -= (1 << s) - 1
            dcs[comp] += v
            pos += ln + s
            k = 1
        else:
            rs, ln = symbol(bits, pos, ac_lut)
            r, s = rs >> 4, rs & 15
            pos += ln + s
            k = k + 16 if (s == 0 and r == 15) else 64 if s == 0 else k + r + 1
        if k >= 64:
            k, blocks, slot = 0, blocks + 1, (slot + 1) % len(slots)
    return (pos, slot, k), blocks, dcs


def synchronise(bits, slots, sub_bits):
    n = -(-len(bits) // sub_bits)
    limit = [min((t + 1) * sub_bits, len(bits)) for t in range(n)]
    used = [(t * sub_bits, 0, 0) for t in range(n)]            # round 0: every subsequence from its first bit, first block of an MCU, DC
    rec = [run(bits, slots, used[t], limit[t]) for t in range(n)]
    rounds = 1
    while True:
        stale = [t for t in range(1, n) if rec[t - 1][0] != used[t]]
        if not stale:
            return rec, used, rounds
        for t in stale:                                           # (all of a round from the exits of the round before, as one kernel launch sees them at worst)
            used[t] = rec[t - 1][0]
        for t in stale:
            rec[t] = run(bits, slots, used[t], limit[t]) if used[t][0] < limit[t] else (used[t], 0, [0, 0, 0])
        rounds += 1
        assert rounds <= n + 1, "more rounds than subsequences"


@pytest.mark.parametrize("name,sub_bits,max_rounds", [("file_q85_420.jpg", 2048, 6), ("file_q85_420.jpg", 512, 20), ("file_q92_444.jpg", 2048, 4),
                                                      ("file_q92_444.jpg", 256, 12), ("file_q80_grey.jpg", 1024, 4), ("file_q88_422.jpg", 128, 24),
                                                      ("file_q85_114.jpg", 2048, 4)])
def test_fixed_point_of_the_rounds_is_the_sequential_decode(name, sub_bits, max_rounds):
    data = open(os.path.join(GOLDEN, name), "rb").read()
    bits, slots, mcus, ncomp = parse(data)
    n = -(-len(bits) // sub_bits)
    # the truth: one decoder from the first bit, its state whenever it crosses a boundary
    truth, state = [], (0, 0, 0)
    for t in range(n):
        exit_, blocks, dcs = run(bits, slots, state, min((t + 1) * sub_bits, len(bits)))
        truth.append((state, exit_, blocks, dcs))
        state = exit_
    assert sum(tr[2] for tr in truth) >= mcus * len(slots), "the fixture's scan holds all its blocks"
    rec, used, rounds = synchronise(bits, slots, sub_bits)
    assert rounds <= max_rounds, (rounds, n)
    for t in range(n):
        assert used[t] == truth[t][0] and rec[t] == truth[t][1:], (name, sub_bits, t)
    # what the scan kernel makes of it: every subsequence's first block and DC predictors are prefix sums over the subsequences in front
    first, pred = 0, [0, 0, 0]
    for t in range(n):
        assert first == sum(tr[2] for tr in truth[:t])
        first += rec[t][1]
        pred = [a + b for a, b in zip(pred, rec[t][2])]
    assert first >= mcus * len(slots)
