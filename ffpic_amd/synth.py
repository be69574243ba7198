"""Synthetic JPEG coefficient grids.

Host-side input generator shared by the tests and bench.py.  It produces what the
entropy decoder of the reference would hand to the reconstruction stage:
quantised coefficients, natural (de-zigzagged) order, int16, blocks in MCU order
per component (index = mcu*(h*v) + vi*h + hi), plus natural-order uint16 quant
tables (format/jpg.c:78-105 de-zigzags DQT at load).

Distribution: quant tables = JPEG Annex K luma/chroma scaled to quality 85 the
libjpeg way; DC ~ N(0, 60) quantised units; AC at zig-zag position k ~ Laplace
with scale 8*exp(-k/6), rounded; everything clipped so |coef * q| <= 2047.
Roughly 85 % of the coefficients are zero.
"""
import numpy as np

SEED_BASE = 0xFF91C

# JPEG Annex K.1 / K.2 example tables, natural (row-major) order.
ANNEX_K_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64)
ANNEX_K_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99], dtype=np.int64)


def zigzag_rank():
    """rank[natural_index] = position of that coefficient in zig-zag scan order."""
    order = sorted(range(64), key=lambda i: (i // 8 + i % 8,
                                             (i // 8) if (i // 8 + i % 8) % 2 else (i % 8)))
    rank = np.empty(64, dtype=np.int64)
    rank[np.array(order)] = np.arange(64)
    return rank


def quant_tables(quality=85):
    """[4][64] uint16, natural order; tables 0/1 = luma/chroma, 2/3 copies."""
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    def one(base):
        return np.clip((base * scale + 50) // 100, 1, 255).astype(np.uint16)
    l, c = one(ANNEX_K_LUMA), one(ANNEX_K_CHROMA)
    return np.stack([l, c, l, c])


def _blocks(rng, n_blocks, q):
    rank = zigzag_rank()
    scale = 8.0 * np.exp(-rank / 6.0)
    ac = np.rint(rng.laplace(0.0, 1.0, size=(n_blocks, 64)) * scale)
    dc = np.rint(rng.normal(0.0, 60.0, size=n_blocks))
    ac[:, 0] = dc
    lim = 2047 // q.astype(np.int64)
    return np.clip(ac, -lim, lim).astype(np.int16)


def coef_image(index, mcu_cols, mcu_rows, ncomp=3, h=2, v=2, qt_id=(0, 1, 1), quant=None):
    """One image: (coef_y, coef_u, coef_v) int16 arrays of shape [blocks, 64]."""
    quant = quant_tables() if quant is None else quant
    rng = np.random.default_rng(SEED_BASE + index)
    mcus = mcu_cols * mcu_rows
    y = _blocks(rng, mcus * h * v, quant[qt_id[0]])
    if ncomp == 1:
        return y, None, None
    u = _blocks(rng, mcus, quant[qt_id[1]])
    w = _blocks(rng, mcus, quant[qt_id[2]])
    return y, u, w


def coef_batch(n_images, mcu_cols, mcu_rows, ncomp=3, h=2, v=2, qt_id=(0, 1, 1), quant=None, first=0):
    """Batch planes, image-major: [n*blocks*64] int16 per component."""
    ys, us, vs = [], [], []
    for i in range(n_images):
        y, u, w = coef_image(first + i, mcu_cols, mcu_rows, ncomp, h, v, qt_id, quant)
        ys.append(y)
        if ncomp == 3:
            us.append(u)
            vs.append(w)
    cat = lambda l: np.ascontiguousarray(np.concatenate(l).reshape(-1)) if l else None
    return cat(ys), cat(us), cat(vs)


def adversarial_blocks(rng, n_blocks):
    """Parity-only inputs that drive the int16 truncation points and the
    mod-2^32 accumulations: full-range int16 levels (use with any quant table)."""
    b = rng.integers(-32768, 32768, size=(n_blocks, 64), dtype=np.int64).astype(np.int16)
    k = n_blocks // 8
    if k:
        b[:k] = 0                                   # all-zero
        b[k:2 * k, 1:] = 0                          # DC only
        b[2 * k:3 * k] = np.where(rng.random((k, 64)) < 0.5, 32767, -32768)   # +-max
        b[3 * k:4 * k, :56] = 0                     # last row only
        m = np.zeros(64, bool); m[7::8] = True
        b[4 * k:5 * k][:, ~m] = 0                   # last column only
    return b


# ------------------------------------------------------------------ VP8 (WebP lossy)

def vp8_quant(segments=4, seed=0):
    """[segments][8] uint16: y1_dc, y1_ac, y2_dc, y2_ac, uv_dc, uv_ac, 0, 0 -- plausible
    values of read_dequantization (format/webp.c:527-543)."""
    rng = np.random.default_rng(SEED_BASE + 7000 + seed)
    q = np.zeros((segments, 8), dtype=np.uint16)
    for s in range(segments):
        base = int(rng.integers(4, 158))
        ac = int(rng.integers(4, 285))
        q[s, :6] = [base, ac, min(2 * base, 132), max(ac * 155 // 100, 8), base, ac]
    return q


def vp8_macroblocks(n_mb, seed=0, adversarial=False):
    """Quantised levels [n_mb][25][16] int16 (raster positions, block 24 = Y2) and the
    per-MB info bytes [n_mb][32]: [0..24] token counts (nz) per block, [25] has_y2,
    [26] segment id.  ~60 % of the MBs carry a Y2 block."""
    rng = np.random.default_rng(SEED_BASE + 9000 + seed)
    if adversarial:
        lv = rng.integers(-32768, 32768, size=(n_mb, 25, 16)).astype(np.int16)
    else:
        scale = 6.0 * np.exp(-np.arange(16) / 4.0)
        lv = np.rint(rng.laplace(0, 1.0, size=(n_mb, 25, 16)) * scale).astype(np.int16)
        lv[rng.random((n_mb, 25)) < 0.35] = 0          # empty blocks
    info = np.zeros((n_mb, 32), dtype=np.uint8)
    # nz as the entropy decoder would report it: index of the last token + 1, in scan
    # order; here a random value consistent with "0 tokens => all zero"
    nzc = (lv != 0).sum(axis=2)
    info[:, :25] = np.where(nzc == 0, 0, np.minimum(16, nzc + rng.integers(0, 3, size=nzc.shape)))
    # the reference quirk: a block whose single token is an AC coefficient (nz == 1, DC == 0)
    k = n_mb // 10
    if k:
        lv[:k, 3, :] = 0
        lv[:k, 3, 5] = 7
        info[:k, 3] = 1
    info[:, 25] = rng.random(n_mb) < 0.6
    info[:, 26] = rng.integers(0, 4, size=n_mb)
    return lv, info


def vp8_modes(mbcols, mbrows, seed=0, bpred_share=0.4):
    """Per-MB mode records [n_mb][20] uint8: intra_y_mode (0 DC, 1 TM, 2 V, 3 H, 4 B_PRED),
    intra_uv_mode (0..3), imodes[16] (0..9), 2 pad bytes."""
    rng = np.random.default_rng(SEED_BASE + 11000 + seed)
    n = mbcols * mbrows
    m = np.zeros((n, 20), dtype=np.uint8)
    m[:, 0] = np.where(rng.random(n) < bpred_share, 4, rng.integers(0, 4, size=n))
    m[:, 1] = rng.integers(0, 4, size=n)
    m[:, 2:18] = rng.integers(0, 10, size=(n, 16))
    return m


def vp8_residual(n_mb, seed=0, amplitude=40):
    rng = np.random.default_rng(SEED_BASE + 12000 + seed)
    r = np.rint(rng.laplace(0, amplitude / 3.0, size=(n_mb, 384))).astype(np.int16)
    r[rng.random(n_mb) < 0.2] = 0
    return r


# ------------------------------------------------------------------ HEVC intra TU lists

HEVC_TU_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2_size", "u1"), ("cidx", "u1"), ("pred_mode", "u1"),
                          ("flags", "u1"), ("res_offset", "<u4"), ("res_scale", "<i4"), ("avail_top", "<u8"),
                          ("avail_left", "<u8")])   # == struct ffhip_hevc_tu, 32 bytes
TU_CORNER, TU_RESIDUAL, TU_FILTER, TU_STRONG, TU_NO_BF, TU_NO_DC_BF, TU_RDPCM, TU_CCP = 1, 2, 4, 8, 16, 32, 64, 128


def _quadtree(rng, x0, y0, size, min_size, max_tu, out, pw=1 << 30, ph=1 << 30):
    """pw, ph: the plane's size -- a block that crosses the picture edge is split (the implicit split of 7.3.8.4), one
    that lies outside it does not exist"""
    if x0 >= pw or y0 >= ph:
        return
    crosses = x0 + size > pw or y0 + size > ph
    if size > max_tu or crosses or (size > min_size and rng.random() < 0.55):
        assert size > min_size or not crosses, "picture size must be a multiple of the smallest TU"
        h = size // 2
        for (dx, dy) in ((0, 0), (h, 0), (0, h), (h, h)):          # z-order
            _quadtree(rng, x0 + dx, y0 + dy, h, min_size, max_tu, out, pw, ph)
    else:
        out.append((x0, y0, size))


def _c5_mix(rng, x0, y0, size, big, out):
    """every `big` x `big` area is one TU with probability 0.6, else four of half the size (z-order)"""
    if size > big:
        h = size // 2
        for (dx, dy) in ((0, 0), (h, 0), (0, h), (h, h)):
            _c5_mix(rng, x0 + dx, y0 + dy, h, big, out)
    elif rng.random() < 0.6:
        out.append((x0, y0, size))
    else:
        h = size // 2
        out.extend(((x0, y0, h), (x0 + h, y0, h), (x0, y0 + h, h), (x0 + h, y0 + h, h)))


def _reference_cu_order(rng, per_plane, cx, cy, ctb, csub):
    """The order decode_cu_coded_intra_prediction_mode (coding/hevc.c:5013-5180) walks one coding tree block in: coding unit by
    coding unit in z-order, and per unit the whole luma transform tree, then the Cb tree, then the Cr tree (one
    decode_intra_block call per component, hevc.c:4665-4805).  per_plane[c] = this block's TUs of plane c as (index, x, y, n) in
    z-order; a coding unit is a quadtree node of 64 / 32 / 16 / 8 luma samples that no TU of any plane straddles (the node is
    split further with probability 0.6 where the TUs allow it).  Returns the indices in that order."""
    out = []

    def inside(c, x0, y0, size):
        sc = 1 if c == 0 else csub
        x0, y0, size = x0 // sc, y0 // sc, size // sc
        return [t for t in per_plane[c] if x0 <= t[1] < x0 + size and y0 <= t[2] < y0 + size]

    def cu(x0, y0, size):
        tus = [inside(c, x0, y0, size) for c in range(len(per_plane))]
        if not any(tus):
            return
        h = size // 2
        can_split = h >= 8 and all(t[3] <= h // (1 if c == 0 else csub) for c in range(len(per_plane)) for t in tus[c])
        if can_split and rng.random() < 0.6:
            for (dx, dy) in ((0, 0), (h, 0), (0, h), (h, h)):
                cu(x0 + dx, y0 + dy, h)
        else:
            for c in range(len(per_plane)):
                out.extend(t[0] for t in tus[c])
    cu(cx, cy, ctb)
    return out


def hevc_reference_order(tus, ctb=64, csub=2, seed=0, return_perm=False):
    """A list as hevc_intra_tus(order="plane") makes it -- per coding tree block the luma TUs, then Cb, then Cr -- in the order the
    reference decodes it in (_reference_cu_order).  Every plane keeps its own order, so availability masks stay what they are."""
    rng = np.random.default_rng(SEED_BASE + 15500 + seed)         # (its own stream: the TUs are the same in both orders)
    sc = np.where(tus["cidx"] == 0, 1, csub).astype(np.int64)
    bx, by = tus["x"].astype(np.int64) * sc // ctb, tus["y"].astype(np.int64) * sc // ctb
    starts = np.concatenate([[0], np.nonzero((bx[1:] != bx[:-1]) | (by[1:] != by[:-1]))[0] + 1, [len(tus)]])
    n_planes = int(tus["cidx"].max()) + 1 if len(tus) else 1
    x, y, n, c = tus["x"].tolist(), tus["y"].tolist(), (1 << tus["log2_size"].astype(np.int64)).tolist(), tus["cidx"].tolist()
    perm = []
    for a, b in zip(starts[:-1].tolist(), starts[1:].tolist()):
        per_plane = [[] for _ in range(n_planes)]
        for i in range(a, b):
            per_plane[c[i]].append((i, x[i], y[i], n[i]))
        got = _reference_cu_order(rng, per_plane, int(bx[a]) * ctb, int(by[a]) * ctb, ctb, csub)
        assert len(got) == b - a
        perm.extend(got)
    perm = np.asarray(perm, dtype=np.int64)
    assert np.array_equal(np.sort(perm), np.arange(len(tus)))
    return (tus[perm], perm) if return_perm else tus[perm]


def hevc_intra_tus(width, height, seed=0, ctb=64, chroma=True, min_tu=4, adversarial_masks=False, ccp=False,
                   chroma_444=False, tu_mix=None, order="plane"):
    """A whole intra picture as a list of TUs in decode order (CTBs in raster order, z-order
    inside a CTB; per CTB: luma TUs, then Cb, then Cr), with z-scan neighbour availability,
    random modes 0..34 and flags.  order="reference": the SAME TUs (records, residuals, availability) in the order the
    reference decodes them in -- per coding unit the luma tree, then Cb, then Cr, with coding units of 64 / 32 / 16 / 8
    inside a coding tree block (_reference_cu_order): every plane keeps its own order, the planes interleave.  ccp=True marks about half of the chroma TUs that carry a residual
    for cross-component prediction (ResScaleVal in {+-1, +-2, +-4, +-8}); chroma_444=True gives the chroma
    planes the luma size (ChromaArrayType 3, where the reference enables it).
    tu_mix="c5" replaces the random quadtree by the TU mix SURVEY 8d names for BASELINE config 5: luma 32/16 at
    60/40, chroma 16/8 at 60/40.
    Returns (tus structured array, residual int16 flat)."""
    rng = np.random.default_rng(SEED_BASE + 15000 + seed)
    assert (width % ctb == 0 and height % ctb == 0) or (tu_mix is None and width % 8 == 0 and height % 8 == 0)
    csub = 1 if chroma_444 else 2
    planes = [(width, height)] + ([(width // csub, height // csub)] * 2 if chroma else [])
    done = [np.zeros((h, w), bool) for (w, h) in planes]
    tus, res_parts, off = [], [], 0
    for cy in range(0, height, ctb):
        for cx in range(0, width, ctb):
            for c, (pw, ph) in enumerate(planes):
                sc = 1 if c == 0 else csub
                parts = []
                if tu_mix == "c5":
                    _c5_mix(rng, cx // sc, cy // sc, ctb // sc, 32 // sc, parts)
                else:
                    _quadtree(rng, cx // sc, cy // sc, ctb // sc, max(min_tu, 4), 32, parts, pw, ph)
                for (x0, y0, n) in parts:
                    d = done[c]
                    at = al = 0
                    for k in range(2 * n):
                        if y0 > 0 and x0 + k < pw and d[y0 - 1, x0 + k]:
                            at |= 1 << k
                        if x0 > 0 and y0 + k < ph and d[y0 + k, x0 - 1]:
                            al |= 1 << k
                    fl = TU_CORNER if (x0 > 0 and y0 > 0 and d[y0 - 1, x0 - 1]) else 0
                    if adversarial_masks and rng.random() < 0.3:       # arbitrary subsets of what exists
                        at &= int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
                        al &= int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 4)) << 62)
                        if rng.random() < 0.5:
                            fl = 0
                    mode = int(rng.integers(0, 35))
                    if rng.random() < 0.2:
                        mode = int(rng.choice([0, 1, 10, 26, 2, 18, 34]))
                    if c == 0 or rng.random() < 0.3:
                        fl |= TU_FILTER
                    fl |= TU_STRONG if rng.random() < 0.7 else 0
                    fl |= TU_NO_BF if rng.random() < 0.15 else 0
                    fl |= TU_NO_DC_BF if rng.random() < 0.15 else 0
                    if rng.random() < 0.75:
                        fl |= TU_RESIDUAL
                        if mode in (10, 26) and rng.random() < 0.5:
                            fl |= TU_RDPCM
                        res_parts.append(np.rint(rng.laplace(0, 12, size=n * n)).astype(np.int16))
                        ro = off
                        off += n * n
                    else:
                        ro = 0
                    rsv = 0
                    if ccp and c > 0 and (fl & TU_RESIDUAL) and rng.random() < 0.5:
                        fl |= TU_CCP
                        rsv = int(rng.choice([1, 2, 4, 8])) * int(rng.choice([-1, 1]))
                    tus.append((x0, y0, int(np.log2(n)), c, mode, fl, ro, rsv, at, al))
                    d[y0:y0 + n, x0:x0 + n] = True
    assert order in ("plane", "reference")
    arr = np.array(tus, dtype=HEVC_TU_DTYPE)
    if order == "reference":
        arr = hevc_reference_order(arr, ctb, csub, seed)
    res = np.concatenate(res_parts) if res_parts else np.zeros(1, np.int16)
    return arr, res


def vp8_filters(seed=0):
    """[4 segments][2 (i16 / i4x4)][3] uint8 = sub_limit, inter_limit, hev_thresh the way
    calculate_filter_control_parameter (format/webp.c:1756-1803) derives them from a level."""
    rng = np.random.default_rng(SEED_BASE + 13000 + seed)
    f = np.zeros((4, 2, 3), dtype=np.uint8)
    sharp = int(rng.integers(0, 8))
    for s in range(4):
        for k in range(2):
            level = int(rng.integers(0, 64)) if rng.random() < 0.9 else 0
            if level > 0:
                il = level
                if sharp > 0:
                    il >>= 2 if sharp > 4 else 1
                    il = min(il, 9 - sharp)
                il = max(il, 1)
                f[s, k] = [(level << 1) + il, il, 2 if level >= 40 else (1 if level >= 15 else 0)]
    return f


def vp8_blocky_planes(mbcols, mbrows, seed=0):
    """Reconstructed-looking 8-bit planes (smooth ramp + per-4x4-block offsets + light noise) on
    which both the simple and the normal VP8 loop filter fire often."""
    rng = np.random.default_rng(SEED_BASE + 14000 + seed)
    out = []
    for (h, w) in ((16 * mbrows, 16 * mbcols), (8 * mbrows, 8 * mbcols), (8 * mbrows, 8 * mbcols)):
        yy, xx = np.mgrid[0:h, 0:w]
        ramp = 60 + 120 * (xx / max(w - 1, 1)) * (yy / max(h - 1, 1)) + 30 * np.sin(xx / 9.0)
        blk = rng.integers(-9, 10, size=(h // 4 + 1, w // 4 + 1))[yy // 4, xx // 4]
        mbo = rng.integers(-14, 15, size=(h // 16 + 1, w // 16 + 1))[yy // 16, xx // 16]
        noise = rng.integers(-1, 2, size=(h, w))
        out.append(np.clip(ramp + blk + mbo + noise, 0, 255).astype(np.uint8))
    return out
