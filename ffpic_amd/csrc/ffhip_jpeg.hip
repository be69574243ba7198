/*
 * ffhip_jpeg.hip -- JPEG post-entropy reconstruction for gfx950 (MI355X).
 *
 * Fused dequant + 8x8 IDCT + YCbCr->BGRA for batches of coefficient grids,
 * bit-exact with the reference's scalar C:
 *   dequant_data_unit      format/jpg.c:247-253
 *   idct_8x8_16            utils/idct.c:512-534 (table :358-367)
 *   YUV_to_BGRA32_16bit    utils/colorspace.c:133-172
 * driven the way the MCU loop of JPG_decode_scan does (format/jpg.c:512-560).
 *
 * Design (see DESIGN.md for the derivations):
 *  - HBM-bound byte/integer work: 3 B of int16 coefficients in, 4 B of BGRA out
 *    per pixel at 4:2:0.  No MFMA; the 8-point transforms are packed-int16 dot
 *    products (v_dot2_i32_i16) on even/odd input pairs, accumulating mod 2^32
 *    exactly like the reference's int arithmetic.
 *  - One wave owns a "quad" = 4 horizontally adjacent MCUs (64x16 pixels) and
 *    works in three rounds of 8 blocks: chroma (4 U + 4 V), luma of MCU 0-1,
 *    luma of MCU 2-3.  Every global load is one dwordx4 per lane (a block row),
 *    1 KiB contiguous per wave; every global store is one dwordx4 per lane with
 *    8 lanes covering one 128-B line of an output row.
 *  - Per round the 8x8 transposes go through a 1 KiB per-wave LDS tile with
 *    gfx950's ds_read_b64_tr_b16 (transposing read), whose row order is chosen
 *    so the lane receives (x0,x4),(x2,x6) / (x1,x3),(x5,x7) pairs ready for the
 *    even/odd butterflies.  Waves never synchronise with each other.
 *  - Colour: R and B have exact integer forms on the IDCT's output domain
 *    [0,8191]; G does too except when 215*uu+381*vv is a non-zero multiple of
 *    1000, where the reference's double roundings decide -- those pixels are
 *    re-evaluated in contraction-free fp64 (tests/tools/check_color_int.c
 *    enumerates the whole domain).  Chroma-only terms are computed once per
 *    chroma sample and shared by its 4 pixels through LDS.
 *  - Geometries other than 3-component h=v=2 take a two-kernel path (IDCT to
 *    int16 sample planes in the workspace, then a per-pixel colour kernel using
 *    the literal fp64 expressions).
 *
 * This file must be compiled with -ffp-contract=off.
 */
#include "ffhip_colorterms.h"

#include <errno.h>
#include <stdlib.h>
#include <mutex>

#ifndef WAVES_PER_WG
#define WAVES_PER_WG 4
#endif
#ifndef FFHIP_LDS_PAD
#define FFHIP_LDS_PAD 0 /* experiment knob: extra LDS per wave to lower occupancy */
#endif
#define FFHIP_JPEG_DEFAULT_VARIANT 13 /* <quads per wave><nt bits> */
#ifndef FFHIP_JPEG_STRIPS_DEFAULT
#define FFHIP_JPEG_STRIPS_DEFAULT 2 /* strips' worth per wave of the 4:4:4, 4:2:2 and 4:4:0 kernels (FFHIP_JPEG_STRIPS=1 / 2 at run time, grey too) */
#endif
#define WG_THREADS (64 * WAVES_PER_WG)

/* per-wave LDS layout (bytes) */
#define LDS_W 0       /* 1 KiB work tile: stage A / B / C and chroma samples */
#define LDS_TR 1024   /* R terms, 64 entries x 16 B                          */
#define LDS_TG 2048   /* G terms                                             */
#define LDS_TB 3072   /* B terms                                             */
#define LDS_UV 4096   /* raw (uu,vv) pairs for the fp64 fallback             */
#define LDS_FL 5120   /* sensitivity masks, 64 x 4 B                         */
#define LDS_WAVE_BYTES (5376 + FFHIP_LDS_PAD)

#define PK16(lo, hi) ((u32)(uint16_t)(int16_t)(lo) | ((u32)(uint16_t)(int16_t)(hi) << 16))

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int dot2(u32 a, u32 b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}

/* One 8-point inverse DCT with the 13-bit basis of utils/idct.c:358-367, inputs
 * as packed int16 pairs e0=(x0,x4) e1=(x2,x6) o0=(x1,x3) o1=(x5,x7); `rnd` is
 * folded into the even part.  All arithmetic mod 2^32 (== the reference's int). */
/* dot2 with a zero accumulator as the 3-source VOP3P form (inline constant 0): hipcc would pick
 * the 2-address v_dot2c and spend a v_mov per chain start to zero its accumulator */
__device__ __forceinline__ int dot2z(u32 a, u32 k)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "s"(k));
    return d;
}

/* 3-address VOP3P form with the accumulator in a VGPR: the even part chains its four outputs off ap / am directly
 * (E0 = ap + b0 and E3 = ap - b0 as two dot2 with negated constants: 6 instructions for the even part instead of 8), and
 * the rounding constant lives in a VGPR for the whole kernel instead of a v_mov per chain start */
__device__ __forceinline__ int dot2a(u32 a, u32 k, int acc)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(acc));
    return d;
}

__device__ __forceinline__ void idct8_1d(u32 e0, u32 e1, u32 o0, u32 o1, int rnd, int out[8])
{
    const int ap = dot2a(e0, PK16(8192, 8192), rnd);
    const int am = dot2a(e0, PK16(8192, -8192), rnd);
    const int E0 = dot2a(e1, PK16(10703, 4433), ap), E3 = dot2a(e1, PK16(-10703, -4433), ap);
    const int E1 = dot2a(e1, PK16(4433, -10704), am), E2 = dot2a(e1, PK16(-4433, 10704), am);
    const int O0 = dot2(o0, PK16(11363, 9633), dot2z(o1, PK16(6437, 2260)));
    const int O1 = dot2(o0, PK16(9633, -2259), dot2z(o1, PK16(-11362, -6436)));
    const int O2 = dot2(o0, PK16(6437, -11362), dot2z(o1, PK16(2261, 9633)));
    const int O3 = dot2(o0, PK16(2260, -6436), dot2z(o1, PK16(9633, -11363)));
    out[0] = E0 + O0; out[7] = E0 - O0;
    out[1] = E1 + O1; out[6] = E1 - O1;
    out[2] = E2 + O2; out[5] = E2 - O2;
    out[3] = E3 + O3; out[4] = E3 - O3;
}

/* low 16 bits of (b >> 11) into the high half of d, low half kept: with a plain shift of the pair's first value before
 * it, two instructions per packed pair instead of two shifts and a byte permute */
__device__ __forceinline__ u32 pack_shr11(int a, int b)
{
    u32 d = (u32)(a >> 11);
    asm("v_ashrrev_i32_sdwa %0, 11, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(d) : "v"(b));
    return d;
}

__device__ __forceinline__ u32 pk_mul16(u32 a, u32 b)
{
    using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b)));
}

/* byte offset of row `u` of block `b` in the 8-block work tile; blocks 2,3,6,7
 * keep their even/odd rows swapped so the transposing reads are conflict-free */
__device__ __forceinline__ u32 tile_off(u32 b, u32 u) { return b * 128u + ((u ^ ((b >> 1) & 1u)) << 4); }

struct WaveCtx {
    char *lds;      /* this wave's LDS region (generic pointer)              */
    u32 lane;
    u32 wr_off;     /* tile_off(block, row) of the row this lane loads        */
    u32 tr_even;    /* byte offsets this lane supplies to the transposing     */
    u32 tr_odd;     /*   reads (even rows 0,4,2,6 / odd rows 1,3,5,7)         */
    u32 blk, idx;   /* after a transposing read: block and column/row index   */
    int rnd1, rnd2; /* the two passes' rounding constants, held in VGPRs               */
};

__device__ __forceinline__ void wave_ctx_init(WaveCtx &c, char *lds, u32 lane)
{
    c.lds = lds;
    c.lane = lane;
    c.wr_off = tile_off(lane >> 3, lane & 7);
    const u32 g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const u32 b = 2 * g + (p >> 1);
    c.tr_even = tile_off(b, ((q & 1) << 2) | (q & 2)) + (p & 1) * 8;
    c.tr_odd = tile_off(b, 2 * q + 1) + (p & 1) * 8;
    c.blk = 2 * g + (t >> 3);
    c.idx = t & 7;
    c.rnd1 = 1 << 10;
    c.rnd2 = 257 << 17;
    asm volatile("" : "+v"(c.rnd1), "+v"(c.rnd2)); /* opaque: otherwise rematerialised at every use */
}

__device__ __forceinline__ u32x2 lds_tr_read(const WaveCtx &c, u32 off)
{
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(c.lds + LDS_W + off));
    return __builtin_bit_cast(u32x2, v);
}

/* Dequantise + 2-D IDCT of 8 blocks held one row per lane (lane = 8*block+row).
 * raw/quant: the lane's 8 coefficients / quant factors as packed int16 pairs.
 * Returns the 8 samples of row c.idx of block c.blk (values 0..8191) as packed
 * int16 pairs (s0,s1),(s2,s3),(s4,s5),(s6,s7). */
__device__ __forceinline__ u32x4 idct8x8_round(const WaveCtx &c, u32x4 raw, u32x4 quant)
{
    /* dequant: low 16 bits of the product = the int16 store of jpg.c:251 */
    u32x4 dq;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 a = raw[i], b = quant[i]; /* scalars first: bit_cast of a vector element lvalue miscompiles */
        dq[i] = pk_mul16(a, b);
    }
    *(u32x4 *)(c.lds + LDS_W + c.wr_off) = dq;                 /* stage A: [block][row u][x] */
    u32x2 ev = lds_tr_read(c, c.tr_even), od = lds_tr_read(c, c.tr_odd);
    int col[8];
    idct8_1d(ev[0], ev[1], od[0], od[1], c.rnd1, col);       /* column x = c.idx, all y */
    /* (v >> 11) stored to int16 (idct.c:522) */
    u32x4 pk;
#pragma unroll
    for (int i = 0; i < 4; i++)
        pk[i] = pack_shr11(col[2 * i], col[2 * i + 1]);
    *(u32x4 *)(c.lds + LDS_W + tile_off(c.blk, c.idx)) = pk;   /* stage B: [block][col x][y] */
    ev = lds_tr_read(c, c.tr_even);
    od = lds_tr_read(c, c.tr_odd);
    int out[8];
    idct8_1d(ev[0], ev[1], od[0], od[1], c.rnd2, out);      /* row y = c.idx, all x */
    /* clamp((v >> 18), 0, 65535): v >> 18 is in [-8192, 8191] so only the lower clamp can
     * act and the int16 store never wraps (idct.c:531).  Done two samples at a time on
     * the high halves: (v >> 16) as int16, >> 2, max 0. */
    u32x4 res;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 hi2 = __builtin_amdgcn_perm((u32)out[2 * i + 1], (u32)out[2 * i], 0x07060302u);
        s16x2 v = __builtin_bit_cast(s16x2, hi2) >> 2;
        v = __builtin_elementwise_max(v, (s16x2){0, 0});
        res[i] = __builtin_bit_cast(u32, v);
    }
    return res;
}

struct JpegBatch {
    const int16_t *coef_y, *coef_u, *coef_v;
    const uint16_t *quant;
    long long quant_stride;
    uint8_t *bgra;
    long long pitch, image_stride;
    int mcu_cols, mcu_rows, quads_per_row, n_images;
    int qt_y, qt_u, qt_v;
    int quads_per_image;   /* quads_per_row * mcu_rows                                  */
    u32 qpr_magic;         /* floor(2^32 / quads_per_row) + 1: row = mulhi(quad, magic) */
    int wgs_per_image;     /* workgroups that cover one image                            */
    u32 wpi_magic;         /* same trick for image = workgroup / wgs_per_image          */
    int xcd_remap;         /* 1: give each XCD a contiguous chunk of the workgroup sequence */
    int pattern_only;      /* host side: launch the arithmetic-free twin (ffhip_jpeg_pattern_calibrate) */
};

/* exact floor((x)/d) for the small non-negative ranges of the chroma terms:
 * (int)((float)(2x+1) * fl(1/(2d))) -- verified exhaustively for d = 25, 125,
 * 1000 over the ranges used below (tests/test_color_forms.py) */
__device__ __forceinline__ int fdiv_f32(int two_x_plus_1, float inv_2d)
{
    return (int)((float)two_x_plus_1 * inv_2d);
}

/* The three chroma terms of one chroma sample pair, from the RAW samples a = uu + 128, b = vv + 128 in [0, 8191], one
 * v_fma_f32 and one v_add_f32 each: the exact quotient sits at least 0.5/d from a rounding boundary ((x + 0.5)/d - 0.5
 * form), and adding 2^23 + 65536 - bias rounds it to the integer and leaves the term's two's-complement int16 in the LOW
 * HALF of the float's bit pattern, ready for the packed 16-bit adds (the high half is garbage).
 *   r: floor(32 vv / 25), g: floor(-(215 uu + 381 vv) / 1000), b: floor(266 uu / 125);
 *   sens: 215 uu + 381 vv is a non-zero multiple of 1000 -- the reference's double roundings decide G there.
 * 12 VALU instructions against 27 for the integer / cvt form it replaces; every a, b enumerated against the integer
 * definitions in tests/tools/check_color_fma.c (0 mismatches of 67 M pairs). */
struct TermBits {
    u32 r, g, b;
    bool sens;
};
__device__ __forceinline__ TermBits chroma_term_bits(u32 a_raw, u32 b_raw)
{
    const float af = (float)a_raw, bf = (float)b_raw;
    TermBits t;
    t.r = __builtin_bit_cast(u32, __builtin_fmaf(bf, 1.28f, -0.32f) + 8453980.0f);
    t.b = __builtin_bit_cast(u32, __builtin_fmaf(af, 2.128f, 0.12f) + 8453871.0f);
    const float sf = __builtin_fmaf(bf, 381.0f, af * 215.0f); /* exact: an integer below 2^24 */
    const float tf = 4882288.0f - sf;                         /* 4806000 - (215 uu + 381 vv) >= 0, exact */
    const float tg = __builtin_fmaf(tf, 0.001f, -0.4995f) + 8449338.0f;
    t.g = __builtin_bit_cast(u32, tg);
    const float kf = tg - 8449338.0f;                         /* floor(tf / 1000), exact */
    t.sens = __builtin_fmaf(kf, -1000.0f, tf) == 0.0f && sf != 76288.0f;
    return t;
}

/* blockIdx.x -> position in the workgroup sequence.  mode 0: as dispatched (round-robin over the XCDs); 1: every XCD
 * gets one contiguous eighth of the sequence; k >= 2: the XCDs take chunks of 2^k workgroups in turn (the part of the grid
 * that is not a whole number of 8 * 2^k stays as dispatched).  Speed only: any placement computes the same bytes. */
__device__ __forceinline__ u32 xcd_remap_wg(int mode)
{
    const u32 nb = gridDim.x, b = blockIdx.x, xcd = b & 7, s = b >> 3;
    if (mode == 0) return b;
    if (mode == 1) {
        const u32 base = nb >> 3, extra = nb & 7;
        return xcd * base + (xcd < extra ? xcd : extra) + s;
    }
    const u32 k = (u32)mode, whole = (nb >> (k + 3)) << (k + 3);
    if (b >= whole) return b;
    return ((s >> k) << (k + 3)) + (xcd << k) + (s & ((1u << k) - 1u));
}

/* ------------------------------------------------------------------------
 * Fused kernel, 3 components, h = v = 2.
 * ---------------------------------------------------------------------- */

/* per-lane, quad-independent roles: everything that depends on the lane only is computed
 * once per wave so the per-quad code addresses LDS as base + compile-time constant */
struct LaneRoles {
    u32 row, lblk;   /* load: block lblk of the round, row `row`                           */
    u32 st_lane;     /* store: 8 lanes = one 128-B piece of an output row                   */
    u32 ce_rd;       /* chroma entry role: e = lane = j*8 + m*2 + hf                        */
    u32 yc_wr;       /* stage-C write offset of the lane's luma row (after pass 2)          */
    u32 yc_rd;       /* stage-C read offset of the lane's 4 output pixels, + k*512          */
    u32 term_rd;     /* term read offset, + k*512 + rnd*64                                  */
    u32 flag_rd;     /* flag read offset, + k*128 + rnd*16                                  */
    u32 flag_sh;     /* bit position of the lane's two flag bits                            */
    u32 lane_m;      /* MCU (0/1) of the round the lane's output pixels belong to           */
};

__device__ __forceinline__ void lane_roles_init(LaneRoles &r, const WaveCtx &c, u32 lane, u32 pitch)
{
    r.row = lane & 7;
    r.lblk = lane >> 3;
    r.st_lane = (lane >> 3) * pitch + (lane & 7) * 16;
    r.ce_rd = (((lane >> 1) & 3) * 8 + (lane >> 3)) * 16 + (lane & 1) * 8;
    {   /* luma row (block c.blk = mloc*4 + vi*2 + hi, row c.idx) -> pixel row, 16-B chunk */
        const u32 mloc = c.blk >> 2, vi = (c.blk >> 1) & 1, hi = c.blk & 1;
        const u32 prow = vi * 8 + c.idx, chunk = mloc * 2 + hi;
        r.yc_wr = prow * 64 + ((chunk ^ ((prow >> 1) & 3)) << 4);
    }
    const u32 cg = lane & 7, l3 = lane >> 3; /* output role: pixel row 8k + l3, 4-pixel group cg */
    r.yc_rd = l3 * 64 + (((cg >> 1) ^ ((l3 >> 1) & 3)) << 4) + (cg & 1) * 8;
    r.term_rd = (l3 >> 1) * 128 + (cg >> 2) * 32 + ((cg >> 1) & 1) * 16 + (cg & 1) * 8;
    r.flag_rd = (l3 >> 1) * 32 + (cg >> 2) * 8 + ((cg >> 1) & 1) * 4;
    r.flag_sh = 2 * (cg & 1);
    r.lane_m = cg >> 2;
}

struct QuadLoads {
    u32x4 c, y0, y1;
};

template <int NT>
__device__ __forceinline__ u32x4 load16(const char *p)
{
    return NT ? __builtin_nontemporal_load((const u32x4 *)p) : *(const u32x4 *)p;
}

/* issue the three 16-B-per-lane loads of one quad (chroma round, luma MCU 0-1, luma MCU 2-3) */
template <int NT>
__device__ __forceinline__ QuadLoads quad_load(const JpegBatch &p, const LaneRoles &r, u32 lane, int img, int mrow,
                                               int mcu0)
{
    const int last = p.mcu_cols - 1;
    const long long mcu_base = ((long long)img * p.mcu_rows + mrow) * p.mcu_cols + mcu0; /* scalar */
    u32 oc = (r.lblk & 3) * 128 + r.row * 16, oy0 = r.lblk * 128 + r.row * 16, oy1 = oy0 + 1024;
    if (__builtin_expect(mcu0 + 3 > last, 0)) { /* ragged right edge (wave-uniform): clamp to the last MCU */
        asm volatile("" ::: "memory");          /* keep this a real branch: the common path pays nothing    */
        const int rem = last - mcu0; /* 0..2 */
        int mc = (int)(r.lblk & 3); mc = mc > rem ? rem : mc;
        int m0 = (int)(r.lblk >> 2), m1 = m0 + 2;
        m0 = m0 > rem ? rem : m0; m1 = m1 > rem ? rem : m1;
        oc = (u32)mc * 128 + r.row * 16;
        oy0 = ((u32)m0 * 4 + (r.lblk & 3)) * 128 + r.row * 16;
        oy1 = ((u32)m1 * 4 + (r.lblk & 3)) * 128 + r.row * 16;
    }
    const char *bu = (const char *)(p.coef_u + mcu_base * 64);
    const char *bv = (const char *)(p.coef_v + mcu_base * 64);
    const char *by = (const char *)(p.coef_y + mcu_base * 256);
    QuadLoads q;
    q.c = load16<NT>((lane < 32 ? bu : bv) + oc);
    q.y0 = load16<NT>(by + oy0);
    q.y1 = load16<NT>(by + oy1);
    return q;
}

/* reconstruct one quad (64x16 pixels) from its loaded coefficients and store it */
template <int NT>
__device__ __forceinline__ void quad_recon(const JpegBatch &p, const WaveCtx &c, const LaneRoles &r, u32 lane,
                                           const QuadLoads &ld, u32x4 q_y, u32x4 q_c, int img, int mrow, int mcu0)
{
    const int last = p.mcu_cols - 1;
    const bool full = mcu0 + 3 <= last; /* scalar */
    /* ---- chroma round: blocks 0-3 = U of MCU 0-3, blocks 4-7 = V ---- */
    {
        const u32x4 pk = idct8x8_round(c, ld.c, q_c);
        /* samples -> work tile [block][row][8 x int16]; then each lane picks up
         * U and V of 4 adjacent chroma columns of one chroma row:
         * entry e = lane = j*8 + m*2 + hf  (chroma row j, MCU m, column half hf) */
        *(u32x4 *)(c.lds + LDS_W + (c.blk * 8 + c.idx) * 16) = pk;
        const u32x2 us = *(const u32x2 *)(c.lds + LDS_W + r.ce_rd);
        const u32x2 vs = *(const u32x2 *)(c.lds + LDS_W + 512 + r.ce_rd);
        u32x4 tr, tg, tb, uv;
        u32 mask = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 uw = us[k >> 1], vw = vs[k >> 1];
            const u32 ua = (k & 1) ? (uw >> 16) : (uw & 0xffffu), va = (k & 1) ? (vw >> 16) : (vw & 0xffffu); /* raw samples */
            const TermBits t = chroma_term_bits(ua, va);
            if (t.sens) mask |= 1u << k; /* exact-integer G: fp64 decides */
            tr[k] = __builtin_amdgcn_perm(t.r, t.r, 0x01000100u);
            tg[k] = __builtin_amdgcn_perm(t.g, t.g, 0x01000100u);
            tb[k] = __builtin_amdgcn_perm(t.b, t.b, 0x01000100u);
            uv[k] = __builtin_amdgcn_perm(va, ua, 0x05040100u); /* raw (u, v) pair for the fp64 fallback: uu = u - 128 (colorspace.c:149) */
        }
        *(u32x4 *)(c.lds + LDS_TR + lane * 16) = tr;
        *(u32x4 *)(c.lds + LDS_TG + lane * 16) = tg;
        *(u32x4 *)(c.lds + LDS_TB + lane * 16) = tb;
        *(u32x4 *)(c.lds + LDS_UV + lane * 16) = uv;
        *(u32 *)(c.lds + LDS_FL + lane * 4) = mask;
    }

    uint8_t *const orow = p.bgra + (long long)img * p.image_stride + (long long)mrow * 16 * p.pitch +
                          (long long)mcu0 * 64; /* scalar */
    /* ---- two luma rounds: MCU 0-1 then MCU 2-3 of the quad ---- */
#pragma unroll
    for (int rnd = 0; rnd < 2; rnd++) {
        /* stage C: 16 pixel rows x 32 px of int16, 64 B per row, 16-B chunks
         * XOR-swizzled by (row>>1)&3 so that both the writes here and the
         * output-order reads below are bank-conflict free */
        {
            const u32x4 pk = idct8x8_round(c, rnd ? ld.y1 : ld.y0, q_y);
            *(u32x4 *)(c.lds + LDS_W + r.yc_wr) = pk;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            /* output role: 8 lanes cover one 32-px row segment (128 B); all LDS addresses are a
             * per-lane base (LaneRoles) plus a compile-time constant */
            const u32x2 yy = *(const u32x2 *)(c.lds + LDS_W + k * 512 + r.yc_rd);
            const u32 toff = k * 512 + rnd * 64 + r.term_rd;
            const u32x2 tr = *(const u32x2 *)(c.lds + LDS_TR + toff);
            const u32x2 tg = *(const u32x2 *)(c.lds + LDS_TG + toff);
            const u32x2 tb = *(const u32x2 *)(c.lds + LDS_TB + toff);
            const u32 fl = (*(const u32 *)(c.lds + LDS_FL + k * 128 + rnd * 16 + r.flag_rd) >> r.flag_sh) & 3u;
            const u32 m = 2 * rnd + r.lane_m;
            u32x4 px;
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const u32 y2 = yy[h2];
                const u32 r2 = sat_pk_u8_i16(pk_add16(y2, tr[h2]));
                const u32 g2 = sat_pk_u8_i16(pk_add16(y2, tg[h2]));
                const u32 b2 = sat_pk_u8_i16(pk_add16(y2, tb[h2]));
                const u32 bg = __builtin_amdgcn_perm(g2, b2, 0x05010400u); /* b0 g0 b1 g1 */
                px[2 * h2] = __builtin_amdgcn_perm(r2, bg, 0x0d040100u);     /* b0 g0 r0 ff */
                px[2 * h2 + 1] = __builtin_amdgcn_perm(r2, bg, 0x0d050302u); /* b1 g1 r1 ff */
            }
            if (fl) { /* rare: exact-integer G decided by the fp64 roundings */
                const u32x2 uvp = *(const u32x2 *)(c.lds + LDS_UV + toff);
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++)
                    if (fl & (1u << h2)) {
                        const u32 uvw = uvp[h2], y2 = yy[h2];
                        const int uu = (int)(uvw & 0xffffu) - 128, vv = (int)(uvw >> 16) - 128;
                        const u32 g0 = green_fp64((int)(y2 & 0xffffu), uu, vv);
                        const u32 g1 = green_fp64((int)(y2 >> 16), uu, vv);
                        px[2 * h2] = (px[2 * h2] & 0xffff00ffu) | (g0 << 8);
                        px[2 * h2 + 1] = (px[2 * h2 + 1] & 0xffff00ffu) | (g1 << 8);
                    }
            }
            if (full || mcu0 + (int)m <= last) {
                u32x4 *dst = (u32x4 *)(orow + (long long)k * 8 * p.pitch + rnd * 128 + r.st_lane);
                if (NT & 2) __builtin_nontemporal_store(px, dst);
                else *dst = px;
            }
        }
    }
}

/* The same quad with the arithmetic taken out (ffhip_jpeg_pattern_calibrate): the three loads as they are, the four store instructions at the
 * addresses, in the order and with the masks of quad_recon, the stored words an XOR of the loaded ones.  What this runs at is the ceiling of the
 * kernel's ACCESS PATTERN on the buffers it is given -- the figure the kernel's own rate is to be read against (DESIGN.md 5 "Round 6"). */
template <int NT>
__device__ __forceinline__ void quad_pattern(const JpegBatch &p, const LaneRoles &r, const QuadLoads &ld, int img, int mrow, int mcu0)
{
    const int last = p.mcu_cols - 1;
    const bool full = mcu0 + 3 <= last;
    uint8_t *const orow = p.bgra + (long long)img * p.image_stride + (long long)mrow * 16 * p.pitch + (long long)mcu0 * 64;
    const u32x4 s = ld.c ^ ld.y0 ^ ld.y1;
    /* FFHIP_JPEG_PATTERN_SLEEP=<n>: the wave idles n x ~0.4 us between its loads and its stores, as long as the real kernel computes (diagnostics: the
     * pattern with the kernel's residency and none of its instruction issue) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = 1; i < p.pattern_only; i++) __builtin_amdgcn_s_sleep(15);
#pragma unroll
    for (int rnd = 0; rnd < 2; rnd++)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const u32 m = 2 * rnd + r.lane_m;
            if (full || mcu0 + (int)m <= last) {
                u32x4 *dst = (u32x4 *)(orow + (long long)k * 8 * p.pitch + rnd * 128 + r.st_lane);
                const u32x4 px = s + (u32)(2 * rnd + k);
                if (NT & 2) __builtin_nontemporal_store(px, dst);
                else *dst = px;
            }
        }
}

/* QPW quads per wave (adjacent in the MCU row), WAVES_PER_WG waves per workgroup;
 * blockIdx = (quad group, MCU row, image).  A short-lived wave issues all its loads up
 * front and never waits on its own earlier stores (vmcnt is in-order), which streams
 * measurably faster on MI355X than a persistent grid-stride loop (tests/tools/membench.hip:
 * 6.2-6.5 TB/s vs 4.7-5.3 TB/s for a 16 B/lane copy).  NT bit 0: non-temporal loads,
 * bit 1: non-temporal stores. */
template <int QPW, int NT, bool PATTERN = false>
__global__ __launch_bounds__(WG_THREADS) void k_jpeg420_fused(JpegBatch p)
{
    __shared__ __attribute__((aligned(16))) char lds_all[WAVES_PER_WG * LDS_WAVE_BYTES];
    const u32 lane = threadIdx.x & 63;
    /* wave-uniform values are forced into SGPRs: hipcc cannot prove that anything
     * derived from threadIdx is uniform and would run all the index math per lane */
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    /* 1-D grid over (image, slot group); a slot is QPW consecutive quads of the image's row-major
     * quad sequence.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), so
     * the linear id is remapped to give every XCD one contiguous chunk of the sequence: the
     * pieces of an output row then come from one XCD back to back instead of from eight at
     * different times (+6 % on the memory-only pattern, tests/tools/membench_jpeg.hip).
     * Speed only: any placement computes the same bytes. */
    u32 wg;
    {
        wg = xcd_remap_wg(p.xcd_remap);
    }
    int img = (int)__umulhi(wg, p.wpi_magic), wgi = (int)wg - img * p.wgs_per_image; /* scalar */
    if (wgi < 0) { img--; wgi += p.wgs_per_image; }
    if (wgi >= p.wgs_per_image) { img++; wgi -= p.wgs_per_image; }
    const int qidx0 = (int)((u32)wgi * WAVES_PER_WG + wave) * QPW;
    if (qidx0 >= p.quads_per_image) return; /* wave-uniform; no barriers anywhere in this kernel */

    WaveCtx c;
    wave_ctx_init(c, lds_all + wave * LDS_WAVE_BYTES, lane);
    LaneRoles r;
    lane_roles_init(r, c, lane, (u32)p.pitch);

    int mrow[QPW], qcol[QPW];
    QuadLoads ld[QPW];
#pragma unroll
    for (int i = 0; i < QPW; i++) {
        int qi = qidx0 + i;
        qi = qi < p.quads_per_image ? qi : p.quads_per_image - 1; /* duplicate load, never stored */
        int mr = (int)__umulhi((u32)qi, p.qpr_magic), qc = qi - mr * p.quads_per_row; /* scalar */
        if (qc < 0) { mr--; qc += p.quads_per_row; }
        if (qc >= p.quads_per_row) { mr++; qc -= p.quads_per_row; } /* quads_per_row == 1: magic saturates */
        mrow[i] = mr;
        qcol[i] = qc;
        ld[i] = quad_load<NT & 1>(p, r, lane, img, mr, qc * 4);
    }
    const uint16_t *qt = p.quant + (long long)img * p.quant_stride;
    const u32x4 q_y = *(const u32x4 *)(qt + p.qt_y * 64 + r.row * 8);
    const u32x4 q_c = *(const u32x4 *)(qt + (lane < 32 ? p.qt_u : p.qt_v) * 64 + r.row * 8);
#pragma unroll
    for (int i = 0; i < QPW; i++)
        if (qidx0 + i < p.quads_per_image) {
            if (PATTERN) quad_pattern<NT>(p, r, ld[i], img, mrow[i], qcol[i] * 4);
            else quad_recon<NT>(p, c, r, lane, ld[i], q_y, q_c, img, mrow[i], qcol[i] * 4);
        }
}

/* ------------------------------------------------------------------------
 * Fused kernel for the other baseline layouts: 4:4:4 (h = v = 1), 4:2:2 (h = 2), 4:4:0 (v = 2),
 * 4:1:1 (h = 4) and its transpose (v = 4), grey.  One wave reconstructs a strip of 512 pixels -- 8 MCUs
 * of 4:4:4 / grey (64x8), 4 MCUs of 4:2:2 (64x8) or 4:4:0 (32x16) -- or, at h * v = 4, of 1024: 4 MCUs of 4:1:1
 * (128x8) or of its transpose (32x32) -- in 1-3 IDCT rounds of 8 blocks (h * v = 4: two luma rounds and ONE chroma
 * round of 4 U + 4 V blocks; with a strip of 2 MCUs, until round 5, that round carried four idle blocks and the
 * transpose's rows were 64-byte runs: 0.69 and 0.60 of the HBM peak where the paired strips reach 0.76 and 0.72), parks
 * the samples as small
 * int16 planes in LDS and converts 4 pixels per lane and pass with the same exact integer forms
 * as the 4:2:0 kernel (fp64 only where the G sum is an exact multiple of 1000).  Same launch shape:
 * short-lived waves, all loads up front, 16-byte non-temporal stores, XCD-contiguous workgroups.
 * ---------------------------------------------------------------------- */
#define SM_YP 1024
#define SM_UP 2048
#define SM_VP 3072
#define SM_WAVE_BYTES 4096

/* TWO (round 6): two strips' worth per wave for the layouts with h * v <= 2 and grey as well -- 16 MCUs of 4:4:4 / grey (128x8), 8 MCUs of 4:2:2 (128x8) or 4:4:0
 * (64x16): 1 024 pixels, four colour passes, the chroma in two rounds (U, V) or four (4:4:4).  4:4:4 and 4:2:2 are the two layouts that stop short of their access
 * pattern on a fast placement (5.9-6.0 of 6.7-6.8 TB/s, profiles/r6_layout_patterns.jsonl) with the most instructions per byte; what a wave spends before its first
 * IDCT round -- index arithmetic, lane roles, quantiser rows -- is paid once per 1 024 pixels instead of per 512. */
template <int H, int V, int NC, int NT, bool PATTERN = false, int TWO = 0> /* PATTERN: the loads and the stores only (ffhip_jpeg_pattern_calibrate) */
__global__ __launch_bounds__(WG_THREADS) void k_jpeg_fused_strip(JpegBatch p)
{
    constexpr int BPM = H * V;                            /* luma blocks per MCU          */
    constexpr int MPS = ((NC == 1 || BPM == 1) ? 8 : 4) * ((TWO && BPM < 4) ? 2 : 1); /* MCUs per strip (of a wave) */
    constexpr int LR = MPS * BPM / 8;                    /* luma rounds: 4:1:1 and its transpose take TWO strips' worth of luma per wave (1 024 pixels), so that
                                                            their one chroma round (4 U + 4 V blocks) has no idle block -- 1.5 rounds per 512 pixels where the
                                                            single strip took 2 -- and the transpose's rows are runs of 128 bytes, not 64 */
    constexpr int PASSES = 2 * LR;                       /* colour passes of 256 pixels */
    constexpr int SW = MPS * 8 * H, SH = 8 * V;          /* strip size in pixels (512; h * v = 4 or TWO: 1024) */
    constexpr int CW = MPS * 8;                          /* chroma samples per strip row */
    constexpr int GPR = SW / 4;                          /* 4-pixel groups per pixel row */
    constexpr int RPP = 64 / GPR;                        /* lane rows per pass */
    static_assert((LR == 1 || LR == 2) && SW * SH == 512 * LR && BPM <= 4 && BPM != 3 && (H == 1 || V == 1), "strip geometry");
    /* the sample planes in the wave's LDS behind the 1 KB work tile: luma SW x SH, then U and V (8 rows of CW) */
    constexpr int CPB = CW * 8 * 2;                      /* bytes of a chroma plane */
    constexpr int YP = SM_YP, UP = YP + SW * SH * 2, VP = UP + ((LR == 1 && CPB < 1024) ? 1024 : CPB);
    constexpr int WAVE_LDS = NC == 3 ? VP + CPB : UP;
    constexpr int WAVE_BYTES = WAVE_LDS <= SM_WAVE_BYTES ? SM_WAVE_BYTES : (WAVE_LDS + 1023) / 1024 * 1024; /* 4 KB as ever; TWO: 4:4:4 7 KB, 4:2:2 / 4:4:0 5 KB */
    static_assert(UP == SM_UP || LR == 2, "plane offsets");
    __shared__ __attribute__((aligned(16))) char lds_all[WAVES_PER_WG * WAVE_BYTES];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32 wg;
    {
        wg = xcd_remap_wg(p.xcd_remap);
    }
    int img = (int)__umulhi(wg, p.wpi_magic), wgi = (int)wg - img * p.wgs_per_image; /* scalar */
    if (wgi < 0) { img--; wgi += p.wgs_per_image; }
    if (wgi >= p.wgs_per_image) { img++; wgi -= p.wgs_per_image; }
    const int sidx = (int)((u32)wgi * WAVES_PER_WG + wave);
    if (sidx >= p.quads_per_image) return; /* wave-uniform; no barriers in this kernel */
    int mrow = (int)__umulhi((u32)sidx, p.qpr_magic), scol = sidx - mrow * p.quads_per_row;
    if (scol < 0) { mrow--; scol += p.quads_per_row; }
    if (scol >= p.quads_per_row) { mrow++; scol -= p.quads_per_row; }
    const int mcu0 = scol * MPS, last = p.mcu_cols - 1;
    const int rem = last - mcu0 < MPS - 1 ? last - mcu0 : MPS - 1; /* MCUs of this strip that exist, minus one */

    WaveCtx c;
    wave_ctx_init(c, lds_all + wave * WAVE_BYTES, lane);
    const u32 row = lane & 7, lblk = lane >> 3;
    const long long mcu_base = ((long long)img * p.mcu_rows + mrow) * p.mcu_cols + mcu0; /* scalar */
    const uint16_t *qt = p.quant + (long long)img * p.quant_stride;

    /* ---- all loads up front: ragged strips re-read their last MCU, its pixels are never stored ---- */
    u32x4 ly[LR], lc0, lc1, lc2, lc3; /* (lc2, lc3: the second halves of U and V where a wave has sixteen MCUs of 4:4:4) */
#pragma unroll
    for (int lr = 0; lr < LR; lr++) {
        int m = ((int)lblk + 8 * lr) / BPM;
        m = m > rem ? rem : m;
        ly[lr] = load16<NT & 1>((const char *)(p.coef_y + (mcu_base + m) * (64 * BPM) + (((int)lblk + 8 * lr) % BPM) * 64 + row * 8));
    }
    const u32x4 q_y = *(const u32x4 *)(qt + p.qt_y * 64 + row * 8);
    u32x4 q_c0 = q_y, q_c1 = q_y;
    if (NC == 3) {
        if (MPS >= 8) { /* rounds of 8 blocks: U, then V (sixteen MCUs: two of each) */
            const int m = (int)lblk > rem ? rem : (int)lblk;
            lc0 = load16<NT & 1>((const char *)(p.coef_u + (mcu_base + m) * 64 + row * 8));
            lc1 = load16<NT & 1>((const char *)(p.coef_v + (mcu_base + m) * 64 + row * 8));
            if (MPS == 16) {
                const int m2 = (int)lblk + 8 > rem ? rem : (int)lblk + 8;
                lc2 = load16<NT & 1>((const char *)(p.coef_u + (mcu_base + m2) * 64 + row * 8));
                lc3 = load16<NT & 1>((const char *)(p.coef_v + (mcu_base + m2) * 64 + row * 8));
            }
            q_c0 = *(const u32x4 *)(qt + p.qt_u * 64 + row * 8);
            q_c1 = *(const u32x4 *)(qt + p.qt_v * 64 + row * 8);
        } else {        /* one round: blocks 0-3 = U of MCU 0-3, blocks 4-7 = V */
            int m = (int)(lblk & 3);
            m = m > rem ? rem : m;
            lc0 = load16<NT & 1>((const char *)((lane < 32 ? p.coef_u : p.coef_v) + (mcu_base + m) * 64 + row * 8));
            q_c0 = *(const u32x4 *)(qt + (lane < 32 ? p.qt_u : p.qt_v) * 64 + row * 8);
        }
    }

    /* ---- IDCT rounds -> sample planes in LDS: luma SH rows x SW, chroma 8 rows x CW (int16).  The 16-byte chunks of a
     * plane row are XOR-swizzled by the row (sw_off) so that the block-row writes here (lanes of one block are 8 rows
     * apart at the same chunk) and the row-segment reads of the colour passes both touch every bank once: laid out
     * plainly, the writes were 4-way bank conflicts (175 M conflict cycles per launch at 4:4:4, profiles/r1_jpeg_geoms_pmc.txt) ---- */
    /* v = 2: the colour passes take rows 2j (pass 0) and 2j + 1 (pass 1) on the same lane, so that the chroma terms the
     * two rows share are computed once; the luma plane keeps the even rows first, then the odd ones, which keeps the
     * four rows a pass reads at once in four different bank quarters */
    auto yrow_pos = [](u32 r) -> u32 { return (r >> 1) + 8 * (r & 1u); };
    auto sw_off = [](u32 row, u32 col, u32 row_samples) -> u32 { /* byte offset of sample (row, col) in a swizzled plane */
        const u32 key = row_samples >= 64 ? (row & 7u) : ((row >> 2) & 3u);
        return row * row_samples * 2 + ((((col >> 3) ^ key) & (row_samples / 8 - 1)) << 4) + (col & 7u) * 2;
    };
    u32x4 pat = {0u, 0u, 0u, 0u}; /* PATTERN: what gets stored -- an XOR of everything the wave loaded */
    if (PATTERN) {
#pragma unroll
        for (int lr = 0; lr < LR; lr++) pat = pat ^ ly[lr];
        if (NC == 3) pat = pat ^ lc0;
        if (NC == 3 && MPS >= 8) pat = pat ^ lc1;
        if (NC == 3 && MPS == 16) pat = pat ^ lc2 ^ lc3;
    }
#pragma unroll
    for (int lr = 0; lr < LR && !PATTERN; lr++) {
        const u32x4 pk = idct8x8_round(c, ly[lr], q_y);
        const u32 gb = c.blk + 8 * lr, m = gb / BPM, sub = gb % BPM;
        const u32 pcol = (m * H + (H > 1 ? sub : 0)) * 8, prow = (V > 1 ? sub : 0) * 8 + c.idx;
        *(u32x4 *)(c.lds + YP + sw_off(V == 2 ? yrow_pos(prow) : prow, pcol, SW)) = pk;
    }
    if (NC == 3 && !PATTERN) {
        if (MPS >= 8) {
            const u32x4 pu = idct8x8_round(c, lc0, q_c0);
            *(u32x4 *)(c.lds + UP + sw_off(c.idx, c.blk * 8, CW)) = pu;
            const u32x4 pv = idct8x8_round(c, lc1, q_c1);
            *(u32x4 *)(c.lds + VP + sw_off(c.idx, c.blk * 8, CW)) = pv;
            if (MPS == 16) {
                const u32x4 pu2 = idct8x8_round(c, lc2, q_c0);
                *(u32x4 *)(c.lds + UP + sw_off(c.idx, (c.blk + 8) * 8, CW)) = pu2;
                const u32x4 pv2 = idct8x8_round(c, lc3, q_c1);
                *(u32x4 *)(c.lds + VP + sw_off(c.idx, (c.blk + 8) * 8, CW)) = pv2;
            }
        } else {
            const u32x4 pc = idct8x8_round(c, lc0, q_c0);
            if (MPS == 4 || (c.blk & 3) < MPS) /* h*v = 4: blocks 2, 3, 6, 7 of the round are repeats of the strip's last MCU */
                *(u32x4 *)(c.lds + (c.blk < 4 ? UP : VP) + sw_off(c.idx, (c.blk & 3) * 8, CW)) = pc;
        }
    }

    /* ---- colour: 2 passes x 4 pixels per lane; 4/H chroma samples serve them.  Same packed form as the 4:2:0 kernel:
     * per pixel PAIR three 16-bit adds, three saturating packs and three byte permutes ---- */
    uint8_t *const obase = p.bgra + (long long)img * p.image_stride + (long long)mrow * SH * p.pitch + (long long)mcu0 * (32 * H);
    TermBits grey_t = {};
    if (NC == 1) grey_t = chroma_term_bits(0u, 0u); /* U = V = 0 planes (jpg.c:501,552-554): uu = vv = -128, never "sensitive" */
    u32 tr2[2], tg2[2], tb2[2], us[2] = {0, 0}, vs[2] = {0, 0}, sens = 0;
    /* per-lane LDS offsets of the two passes, computed once: pass 1 reads 64 / GPR rows (v = 2: 8 row positions) further
     * on, which flips one bit of the swizzle key -- an XOR and an add instead of a second address computation */
    const u32 row0 = V >= 2 ? 2 * (lane / GPR) : lane / GPR, pc0 = (lane % GPR) * 4;
    const u32 y_off0 = sw_off(V == 2 ? yrow_pos(row0) : row0, pc0, SW);
    /* v = 4: rows 2j and 2j + 1 share (row >> 2), i.e. the swizzle key: the next plane row, 32 bytes on */
    const u32 y_off1 = SW == 64 ? (y_off0 ^ 0x40u) + 4 * 128 : (SW == 32 ? (y_off0 ^ 0x20u) + 8 * 64 : y_off0 + 32);
    static_assert((SW == 64 && 64 / GPR == 4 && V == 1) || (SW == 32 && V == 2) || LR == 2, "pass-1 offset identities");
    /* the pixel row of pass `it`.  v = 1: RPP lane rows a pass, one below the other (128 x 8 pixels: two rows a pass, four passes).  v >= 2: a lane takes rows
     * 2j and 2j + 1 in two passes running (they share their chroma row), RPP such pairs a pass pair -- the transpose of 4:1:1 (32 x 32): the upper half, then the
     * lower half; two strips of 4:4:0 (64 x 16): rows 0-7, then 8-15 */
    auto pass_row = [&](int it) -> u32 {
        return V >= 2 ? row0 + (u32)(it & 1) + (u32)(2 * RPP * (it >> 1)) : row0 + (u32)(it * RPP);
    };
    u32 y_offs[PASSES], c_offs[PASSES];
#pragma unroll
    for (int it = 0; it < PASSES; it++) {
        y_offs[it] = LR == 2 ? sw_off(V == 2 ? yrow_pos(pass_row(it)) : pass_row(it), pc0, SW) : (it ? y_off1 : y_off0); /* (four passes: worked out pass by pass) */
        c_offs[it] = 0;
    }
    const u32 c_off0 = NC == 3 ? sw_off(row0 / V, pc0 / H, CW) : 0;
    /* v = 1: pass 1 is four rows down -- one bit of the key flips and four chroma rows (CW samples each) are skipped */
    const u32 c_off1 = V >= 2 ? c_off0 : (CW == 64 ? (c_off0 ^ 0x40u) + 4 * 128 : (c_off0 ^ 0x10u) + 4 * CW * 2);
#pragma unroll
    for (int it = 0; it < PASSES; it++) c_offs[it] = (LR == 2 && NC == 3) ? sw_off(pass_row(it) / V, pc0 / H, CW) : (it ? c_off1 : c_off0);
#pragma unroll
    for (int it = 0; it < PASSES; it++) {
        const u32 prow = pass_row(it);
        if (PATTERN) { /* the pass's store, at its address and under its mask */
            if (mcu0 + (int)(pc0 / (8 * H)) <= last) {
                u32x4 *dst = (u32x4 *)(obase + (long long)prow * p.pitch + pc0 * 4);
                if (NT & 2) __builtin_nontemporal_store(pat + (u32)it, dst);
                else *dst = pat + (u32)it;
            }
            continue;
        }
        const u32x2 yy = *(const u32x2 *)(c.lds + YP + y_offs[it]);
        if (V >= 2 && (it & 1)) {
            /* the terms of pass 0 serve this row too */
        } else if (NC == 1) {
            tr2[0] = tr2[1] = __builtin_amdgcn_perm(grey_t.r, grey_t.r, 0x01000100u);
            tg2[0] = tg2[1] = __builtin_amdgcn_perm(grey_t.g, grey_t.g, 0x01000100u);
            tb2[0] = tb2[1] = __builtin_amdgcn_perm(grey_t.b, grey_t.b, 0x01000100u);
        } else {
            const u32 c_off = c_offs[it];
            if (H == 1) {
                const u32x2 a = *(const u32x2 *)(c.lds + UP + c_off), b = *(const u32x2 *)(c.lds + VP + c_off);
                us[0] = a[0]; us[1] = a[1]; vs[0] = b[0]; vs[1] = b[1];
            } else if (H == 2) {
                us[0] = *(const u32 *)(c.lds + UP + c_off);
                vs[0] = *(const u32 *)(c.lds + VP + c_off);
                us[1] = vs[1] = 0;
            } else { /* h = 4: the lane's four pixels share one chroma sample */
                us[0] = *(const uint16_t *)(c.lds + UP + c_off);
                vs[0] = *(const uint16_t *)(c.lds + VP + c_off);
                us[1] = vs[1] = 0;
            }
            sens = 0;
            if (H == 4) {
                const TermBits t = chroma_term_bits(us[0], vs[0]);
                sens = t.sens ? 1u : 0u;
                tr2[0] = tr2[1] = __builtin_amdgcn_perm(t.r, t.r, 0x01000100u);
                tg2[0] = tg2[1] = __builtin_amdgcn_perm(t.g, t.g, 0x01000100u);
                tb2[0] = tb2[1] = __builtin_amdgcn_perm(t.b, t.b, 0x01000100u);
            } else {
                constexpr int NP = H == 1 ? 2 : 1; /* sample pairs: two at h = 1, one at h = 2 */
                TermBits2 t[NP];
#pragma unroll
                for (int k = 0; k < NP; k++) t[k] = chroma_term_bits2(us[k], vs[k]);
                float any;
                if (H == 1) {
                    const f32x2 m = t[0].rem * t[NP - 1].rem;
                    any = m.x * m.y;
#pragma unroll
                    for (int h2 = 0; h2 < 2; h2++) { /* one chroma sample per pixel */
                        tr2[h2] = t[h2 % NP].r;
                        tg2[h2] = t[h2 % NP].g;
                        tb2[h2] = t[h2 % NP].b;
                    }
                } else {
                    any = t[0].rem.x * t[0].rem.y;
                    tr2[0] = __builtin_amdgcn_perm(t[0].r, t[0].r, 0x01000100u); /* a pixel pair shares its chroma sample */
                    tr2[1] = __builtin_amdgcn_perm(t[0].r, t[0].r, 0x03020302u);
                    tg2[0] = __builtin_amdgcn_perm(t[0].g, t[0].g, 0x01000100u);
                    tg2[1] = __builtin_amdgcn_perm(t[0].g, t[0].g, 0x03020302u);
                    tb2[0] = __builtin_amdgcn_perm(t[0].b, t[0].b, 0x01000100u);
                    tb2[1] = __builtin_amdgcn_perm(t[0].b, t[0].b, 0x03020302u);
                }
                if (any == 0.0f) { /* rare: some sample's G sum is a multiple of 1000 (zero included) */
#pragma unroll
                    for (int k = 0; k < 2 * NP; k++) {
                        const float rem = (k & 1) ? t[k >> 1].rem.y : t[k >> 1].rem.x, sf = (k & 1) ? t[k >> 1].sf.y : t[k >> 1].sf.x;
                        sens |= (rem == 0.0f && sf != 76288.0f) ? 1u << k : 0u;
                    }
                }
            }
        }
        u32x4 px;
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            const u32 y2 = yy[h2];
            const u32 r2 = sat_pk_u8_i16(pk_add16(y2, tr2[h2]));
            const u32 g2 = sat_pk_u8_i16(pk_add16(y2, tg2[h2]));
            const u32 b2 = sat_pk_u8_i16(pk_add16(y2, tb2[h2]));
            const u32 bg = __builtin_amdgcn_perm(g2, b2, 0x05010400u); /* b0 g0 b1 g1 */
            px[2 * h2] = __builtin_amdgcn_perm(r2, bg, 0x0d040100u);     /* b0 g0 r0 ff */
            px[2 * h2 + 1] = __builtin_amdgcn_perm(r2, bg, 0x0d050302u); /* b1 g1 r1 ff */
        }
        if (NC == 3 && sens) { /* rare: exact-integer G decided by the fp64 roundings */
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int k = d / H;
                if (sens & (1u << k)) {
                    const int y1 = (int)((d & 1) ? (yy[d >> 1] >> 16) : (yy[d >> 1] & 0xffffu));
                    const u32 ua = (k & 1) ? (us[k >> 1] >> 16) : (us[k >> 1] & 0xffffu); /* raw samples: uu = u - 128 (colorspace.c:149) */
                    const u32 va = (k & 1) ? (vs[k >> 1] >> 16) : (vs[k >> 1] & 0xffffu);
                    px[d] = (px[d] & 0xffff00ffu) | (green_fp64(y1, (int)ua - 128, (int)va - 128) << 8);
                }
            }
        }
        if (mcu0 + (int)(pc0 / (8 * H)) <= last) {
            u32x4 *dst = (u32x4 *)(obase + (long long)prow * p.pitch + pc0 * 4);
            if (NT & 2) __builtin_nontemporal_store(px, dst);
            else *dst = px;
        }
    }
}

/* ------------------------------------------------------------------------
 * Generic path, kernel 1: dequant + IDCT of one component plane into int16
 * sample planes (same block-major layout as the input).
 * ---------------------------------------------------------------------- */
struct IdctPlanes {
    const int16_t *coef;
    int16_t *samples;
    const uint16_t *quant;
    long long quant_stride;
    long long blocks_per_image, total_blocks;
    int qt;
};

__global__ __launch_bounds__(WG_THREADS) void k_jpeg_idct_planes(IdctPlanes p)
{
    __shared__ __attribute__((aligned(16))) char lds_all[WAVES_PER_WG * 1024];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveCtx c;
    wave_ctx_init(c, lds_all + wave * 1024, lane);
    const long long n_rounds = (p.total_blocks + 7) / 8;
    const long long n_waves = (long long)gridDim.x * WAVES_PER_WG;
    for (long long r = (long long)blockIdx.x * WAVES_PER_WG + wave; r < n_rounds; r += n_waves) {
        long long b = r * 8 + (lane >> 3);
        b = b < p.total_blocks ? b : p.total_blocks - 1;
        const long long img = b / p.blocks_per_image;
        const u32x4 quant = *(const u32x4 *)(p.quant + img * p.quant_stride + p.qt * 64 + (lane & 7) * 8);
        const u32x4 raw = *(const u32x4 *)(p.coef + b * 64 + (lane & 7) * 8);
        const u32x4 pk = idct8x8_round(c, raw, quant);
        const long long ob = r * 8 + c.blk;
        if (ob < p.total_blocks) *(u32x4 *)(p.samples + ob * 64 + c.idx * 8) = pk;
    }
}

/* Generic path, kernel 2: per-pixel colour conversion from sample planes with the
 * literal expressions of colorspace.c:148-164; one lane = 4 horizontal pixels. */
struct ColorGeneric {
    const int16_t *sy, *su, *sv; /* su/sv NULL for grey: U = V = 0 (jpg.c:501,552-554) */
    uint8_t *bgra;
    long long pitch, image_stride;
    int mcu_cols, mcu_rows, h, v, n_images;
};

__global__ __launch_bounds__(256) void k_jpeg_color_generic(ColorGeneric p)
{
    const int w4 = p.mcu_cols * 8 * p.h / 4, hgt = p.mcu_rows * 8 * p.v;
    const long long per_image = (long long)w4 * hgt, total = per_image * p.n_images;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(t / per_image);
        const int rem = (int)(t - (long long)img * per_image);
        const int y = rem / w4, x = (rem - y * w4) * 4;
        const int my = y / (8 * p.v), i = y - my * 8 * p.v;
        const int mx = x / (8 * p.h), k0 = x - mx * 8 * p.h;
        const long long mcu = ((long long)img * p.mcu_rows + my) * p.mcu_cols + mx;
        const int16_t *Y = p.sy + mcu * p.h * p.v * 64;
        u32x4 px;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int k = k0 + d;
            const int16_t yy = Y[((i / 8) * p.h + (k / 8)) * 64 + (i % 8) * 8 + (k % 8)];
            int su = 0, sv = 0;
            if (p.su) {
                su = p.su[mcu * 64 + (i / p.v) * 8 + (k / p.h)];
                sv = p.sv[mcu * 64 + (i / p.v) * 8 + (k / p.h)];
            }
            const int16_t uu = (int16_t)(su - 128), vv = (int16_t)(sv - 128);
            double dr = (double)yy + 1.280 * (double)vv;
            double dg = (double)yy - 0.215 * (double)uu;
            dg = dg - 0.381 * (double)vv;
            double db = (double)yy + 2.128 * (double)uu;
            int r = (int)dr, g = (int)dg, b = (int)db;
            r = r < 0 ? 0 : (r > 255 ? 255 : r);
            g = g < 0 ? 0 : (g > 255 ? 255 : g);
            b = b < 0 ? 0 : (b > 255 ? 255 : b);
            px[d] = (u32)b | ((u32)g << 8) | ((u32)r << 16) | 0xff000000u;
        }
        *(u32x4 *)(p.bgra + (long long)img * p.image_stride + (long long)y * p.pitch + (long long)x * 4) = px;
    }
}

/* ------------------------------------------------------------------------ host */

static int geom_ok(const ffhip_jpeg_geom *g)
{
    if (!g || g->mcu_cols <= 0 || g->mcu_rows <= 0) return 0;
    if (g->ncomp != 1 && g->ncomp != 3) return 0;
    /* what the reference's MCU loop admits: its scratch is Y[3][64*4] (jpg.c:501) and YUV_to_BGRA32_16bit takes any
     * (v, h) (colorspace.c:143-150) -- every luma sampling pair with h*v <= 4 data units, 4:1:1 and its transpose included */
    if (g->h < 1 || g->v < 1 || g->h * g->v > 4) return 0;
    for (int c = 0; c < g->ncomp; c++)
        if (g->qt_id[c] < 0 || g->qt_id[c] > 3) return 0;
    return 1;
}

/* kernel variant: quads per wave and cache policy.  FFHIP_JPEG_VARIANT="<qpw><nt>" (e.g. "21")
 * overrides the default for experiments; every variant computes identical bytes. */
/* A/B knobs; identical bytes either way.  FFHIP_JPEG_XCD_CHUNK_LOG2=k (2..20): the XCDs take chunks of 2^k workgroups in turn */
static int jpeg_remap_mode(void)
{
    if (FFHIP_ENV("FFHIP_JPEG_NO_XCD_REMAP")) return 0;
    const char *e = FFHIP_ENV("FFHIP_JPEG_XCD_CHUNK_LOG2");
    if (e) { const int k = atoi(e); if (k >= 2 && k <= 20) return k; }
    return 1;
}
static void launch_fused(const JpegBatch &q_in, int n_images, hipStream_t st)
{
    const char *e = FFHIP_ENV("FFHIP_JPEG_VARIANT");
    const int g_variant = (e && e[0] >= '1' && e[0] <= '2' && e[1] >= '0' && e[1] <= '3') ? (e[0] - '0') * 10 + (e[1] - '0') : FFHIP_JPEG_DEFAULT_VARIANT;
    const int qpw = g_variant / 10;
    JpegBatch q = q_in;
    q.xcd_remap = jpeg_remap_mode();
    const int slots = (q.quads_per_image + qpw - 1) / qpw;
    q.wgs_per_image = (slots + WAVES_PER_WG - 1) / WAVES_PER_WG;
    q.wpi_magic = q.wgs_per_image == 1 ? 0xffffffffu : (u32)(0x100000000ULL / (unsigned)q.wgs_per_image) + 1u;
    const dim3 grid((unsigned)((long long)q.wgs_per_image * n_images), 1, 1);
    /* FFHIP_JPEG_LDS_PAD=<bytes>: that much dynamic LDS per workgroup on top of the kernel's own 21 KB, which nobody touches -- fewer workgroups per CU
     * (7 as shipped; 8192 -> 5, 16384 -> 4, 32768 -> 3, 61440 -> 1): the occupancy experiment of DESIGN.md 5 as a run-time switch */
    const char *pe = FFHIP_ENV("FFHIP_JPEG_LDS_PAD");
    const unsigned pad = pe && atoi(pe) > 0 && atoi(pe) <= 120 * 1024 ? (unsigned)atoi(pe) & ~15u : 0u;
    if (q.pattern_only) { /* the variant's twin (non-temporal loads and stores): same grid, same loads, same stores */
        if (qpw == 2) hipLaunchKernelGGL((k_jpeg420_fused<2, 3, true>), grid, dim3(WG_THREADS), pad, st, q);
        else hipLaunchKernelGGL((k_jpeg420_fused<1, 3, true>), grid, dim3(WG_THREADS), pad, st, q);
        return;
    }
#define FFHIP_LAUNCH(Q, N) hipLaunchKernelGGL((k_jpeg420_fused<Q, N>), grid, dim3(WG_THREADS), pad, st, q)
    switch (g_variant) {
    case 10: FFHIP_LAUNCH(1, 0); break;
    case 11: FFHIP_LAUNCH(1, 1); break;
    case 12: FFHIP_LAUNCH(1, 2); break;
    case 13: FFHIP_LAUNCH(1, 3); break;
    case 20: FFHIP_LAUNCH(2, 0); break;
    case 21: FFHIP_LAUNCH(2, 1); break;
    case 22: FFHIP_LAUNCH(2, 2); break;
    default: FFHIP_LAUNCH(2, 3); break;
    }
#undef FFHIP_LAUNCH
}

static int is_fused420(const ffhip_jpeg_geom *g) { return g->ncomp == 3 && g->h == 2 && g->v == 2; }
/* 4:4:4, 4:2:2, 4:4:0, 4:1:1 (h = 4) and its transpose (v = 4), grey: k_jpeg_fused_strip.  The three-block pairs
 * (h or v = 3) and grey with several blocks per MCU -- layouts the reference's loop admits and no encoder writes --
 * take the two-pass path */
static int is_fused_strip(const ffhip_jpeg_geom *g)
{
    return (g->ncomp == 3 && (g->h * g->v <= 2 || g->h == 4 || g->v == 4)) || (g->ncomp == 1 && g->h == 1 && g->v == 1);
}

/* two strips' worth per wave for the layouts with h * v <= 2 and grey (k_jpeg_fused_strip<..., TWO = 1>); h * v = 4 always has.  FFHIP_JPEG_STRIPS=1 / 2 forces either */
static bool strip_two(const ffhip_jpeg_geom *g)
{
    if (g->ncomp == 3 && g->h * g->v == 4) return false;
    const char *e = FFHIP_ENV("FFHIP_JPEG_STRIPS");
    if (e && (e[0] == '1' || e[0] == '2')) return e[0] == '2';
    /* 4:4:4, 4:2:2, 4:4:0: + 2-3 %, + 4-5 %, + 3-6 % on the same output buffer, slow and fast placements alike; grey, which runs at its access pattern with one
     * strip, loses 1-4 % with two (the pattern of 128 x 8-pixel strips is that much slower than that of 64 x 8: profiles/r6_strips_ab*.jsonl) */
    return FFHIP_JPEG_STRIPS_DEFAULT == 2 && g->ncomp == 3;
}
static void launch_strip(const ffhip_jpeg_geom *g, const JpegBatch &q_in, int n_images, hipStream_t st)
{
    JpegBatch q = q_in;
    q.xcd_remap = jpeg_remap_mode();
    q.wgs_per_image = (q.quads_per_image + WAVES_PER_WG - 1) / WAVES_PER_WG;
    q.wpi_magic = q.wgs_per_image == 1 ? 0xffffffffu : (u32)(0x100000000ULL / (unsigned)q.wgs_per_image) + 1u;
    const dim3 grid((unsigned)((long long)q.wgs_per_image * n_images), 1, 1);
    const bool two = strip_two(g);
#define STRIP_LAUNCH(H_, V_, NC_) do { \
        if (q.pattern_only) { /* the arithmetic-free twins: same grids, loads and stores */ \
            if (two) hipLaunchKernelGGL((k_jpeg_fused_strip<H_, V_, NC_, 3, true, 1>), grid, dim3(WG_THREADS), 0, st, q); \
            else hipLaunchKernelGGL((k_jpeg_fused_strip<H_, V_, NC_, 3, true, 0>), grid, dim3(WG_THREADS), 0, st, q); \
        } else if (two) hipLaunchKernelGGL((k_jpeg_fused_strip<H_, V_, NC_, 3, false, 1>), grid, dim3(WG_THREADS), 0, st, q); \
        else hipLaunchKernelGGL((k_jpeg_fused_strip<H_, V_, NC_, 3, false, 0>), grid, dim3(WG_THREADS), 0, st, q); \
    } while (0)
    if (g->ncomp == 1) STRIP_LAUNCH(1, 1, 1);
    else if (g->h == 1 && g->v == 1) STRIP_LAUNCH(1, 1, 3);
    else if (g->h == 2) STRIP_LAUNCH(2, 1, 3);
    else if (g->v == 2) STRIP_LAUNCH(1, 2, 3);
    else if (g->h == 4) STRIP_LAUNCH(4, 1, 3);
    else STRIP_LAUNCH(1, 4, 3);
#undef STRIP_LAUNCH
}

static int grid_for(long long work_items_per_wg_unit)
{
    /* persistent-style grid: enough workgroups to fill 256 CUs several times over,
     * never more than there is work */
    long long want = 256LL * 7;
    if (work_items_per_wg_unit < want) want = work_items_per_wg_unit;
    return (int)(want < 1 ? 1 : want);
}

extern "C" int ffhip_bgra_layout(const ffhip_jpeg_geom *g, int64_t *pitch, int64_t *image_stride)
{
    if (!g || !pitch || !image_stride || !geom_ok(g)) return FFHIP_EINVAL;
    const int64_t w = 8LL * g->h * g->mcu_cols, h = 8LL * g->v * g->mcu_rows;
    *pitch = 4 * w + 1024;
    *image_stride = *pitch * h;
    return FFHIP_OK;
}

extern "C" size_t ffhip_jpeg_workspace_bytes(const ffhip_jpeg_geom *g, int n_images)
{
    if (!geom_ok(g) || n_images <= 0 || is_fused420(g) || is_fused_strip(g)) return 0;
    size_t mcus = (size_t)g->mcu_cols * g->mcu_rows * (size_t)n_images;
    size_t blocks = mcus * (size_t)(g->h * g->v) + (g->ncomp == 3 ? 2 * mcus : 0);
    return blocks * 64 * sizeof(int16_t);
}

extern "C" const char *ffhip_jpeg_kernel_name(const ffhip_jpeg_geom *g)
{
    if (!geom_ok(g)) return "";
    return is_fused420(g) ? "k_jpeg420_fused" : (is_fused_strip(g) ? "k_jpeg_fused_strip" : "k_jpeg_idct_planes");
}

static int jpeg_recon_batch_impl(const ffhip_jpeg_geom *g, int n_images, const int16_t *d_coef_y,
                                 const int16_t *d_coef_u, const int16_t *d_coef_v,
                                 const uint16_t *d_quant, int64_t quant_stride, uint8_t *d_bgra,
                                 int64_t pitch, int64_t image_stride, void *d_workspace,
                                 size_t workspace_bytes, void *stream, const bool pattern_only)
{
    if (!geom_ok(g) || n_images < 0) return FFHIP_EINVAL;
    if (n_images == 0) return FFHIP_OK;
    const int64_t width = (int64_t)g->mcu_cols * 8 * g->h, height = (int64_t)g->mcu_rows * 8 * g->v;
    if (!d_coef_y || !d_quant || !d_bgra) return FFHIP_EINVAL;
    if (g->ncomp == 3 && (!d_coef_u || !d_coef_v)) return FFHIP_EINVAL;
    if (pitch < width * 4 || (pitch & 15) || ((uintptr_t)d_bgra & 15) || (image_stride & 15)) return FFHIP_EINVAL;
    if (n_images > 1 && image_stride < pitch * height) return FFHIP_EINVAL;
    if (quant_stride != 0 && quant_stride < 256) return FFHIP_EINVAL;
    if (((uintptr_t)d_coef_y & 15) || ((uintptr_t)d_coef_u & 15) || ((uintptr_t)d_coef_v & 15) ||
        ((uintptr_t)d_quant & 15) || (quant_stride & 7))
        return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    hipStream_t st = (hipStream_t)stream;

    if (pattern_only && !is_fused420(g) && !is_fused_strip(g)) return FFHIP_EINVAL;
    if (is_fused420(g)) {
        JpegBatch p;
        p.pattern_only = 0;
        if (pattern_only) {
            const char *sl = FFHIP_ENV("FFHIP_JPEG_PATTERN_SLEEP");
            p.pattern_only = 1 + (sl && atoi(sl) > 0 && atoi(sl) < 4096 ? atoi(sl) : 0);
        }
        p.coef_y = d_coef_y; p.coef_u = d_coef_u; p.coef_v = d_coef_v;
        p.quant = d_quant; p.quant_stride = quant_stride;
        p.bgra = d_bgra; p.pitch = pitch; p.image_stride = image_stride;
        p.mcu_cols = g->mcu_cols; p.mcu_rows = g->mcu_rows;
        p.quads_per_row = (g->mcu_cols + 3) / 4; p.n_images = n_images;
        p.quads_per_image = p.quads_per_row * g->mcu_rows;
        p.qpr_magic = p.quads_per_row == 1 ? 0xffffffffu : (u32)(0x100000000ULL / (unsigned)p.quads_per_row) + 1u;
        p.qt_y = g->qt_id[0]; p.qt_u = g->qt_id[1]; p.qt_v = g->qt_id[2];
        long long quads = (long long)p.quads_per_row * p.mcu_rows * n_images;
        if (quads > 0x7fffffffLL || pitch * 16 > 0x7fffffffLL || p.quads_per_image > (1 << 20) || p.quads_per_row > 4096)
            return FFHIP_EINVAL;
        const int max_imgs = (int)(0x7fffffffLL / ((p.quads_per_image + WAVES_PER_WG - 1) / WAVES_PER_WG));
        for (int first = 0; first < n_images; first += max_imgs) {
            const int cnt = n_images - first < max_imgs ? n_images - first : max_imgs;
            JpegBatch q = p;
            const long long mcus = (long long)g->mcu_cols * g->mcu_rows;
            q.coef_y += (long long)first * mcus * 256;
            q.coef_u += (long long)first * mcus * 64;
            q.coef_v += (long long)first * mcus * 64;
            q.quant += (long long)first * quant_stride;
            q.bgra += (long long)first * image_stride;
            q.n_images = cnt;
            launch_fused(q, cnt, st);
            FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
        }
        return FFHIP_OK;
    }

    if (is_fused_strip(g)) {
        const int mps = ((g->ncomp == 1 || g->h * g->v == 1) ? 8 : 4) * (strip_two(g) ? 2 : 1), bpm = g->ncomp == 1 ? 1 : g->h * g->v;
        JpegBatch p = {};
        p.pattern_only = pattern_only ? 1 : 0;
        p.coef_y = d_coef_y; p.coef_u = d_coef_u; p.coef_v = d_coef_v;
        p.quant = d_quant; p.quant_stride = quant_stride;
        p.bgra = d_bgra; p.pitch = pitch; p.image_stride = image_stride;
        p.mcu_cols = g->mcu_cols; p.mcu_rows = g->mcu_rows;
        p.quads_per_row = (g->mcu_cols + mps - 1) / mps; p.n_images = n_images;   /* "quad" = strip here */
        p.quads_per_image = p.quads_per_row * g->mcu_rows;
        p.qpr_magic = p.quads_per_row == 1 ? 0xffffffffu : (u32)(0x100000000ULL / (unsigned)p.quads_per_row) + 1u;
        p.qt_y = g->qt_id[0]; p.qt_u = g->qt_id[1]; p.qt_v = g->qt_id[2];
        const long long strips = (long long)p.quads_per_row * p.mcu_rows * n_images;
        if (strips > 0x7fffffffLL || pitch * 16 > 0x7fffffffLL || p.quads_per_image > (1 << 20) || p.quads_per_row > 4096)
            return FFHIP_EINVAL;
        const int max_imgs = (int)(0x7fffffffLL / ((p.quads_per_image + WAVES_PER_WG - 1) / WAVES_PER_WG));
        for (int first = 0; first < n_images; first += max_imgs) {
            const int cnt = n_images - first < max_imgs ? n_images - first : max_imgs;
            JpegBatch q = p;
            const long long mcus = (long long)g->mcu_cols * g->mcu_rows;
            q.coef_y += (long long)first * mcus * 64 * bpm;
            if (g->ncomp == 3) { q.coef_u += (long long)first * mcus * 64; q.coef_v += (long long)first * mcus * 64; }
            q.quant += (long long)first * quant_stride;
            q.bgra += (long long)first * image_stride;
            q.n_images = cnt;
            launch_strip(g, q, cnt, st);
            FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
        }
        return FFHIP_OK;
    }

    /* remaining geometries (grey with h*v > 1, h or v = 3): IDCT to sample planes, then colour */
    const size_t need = ffhip_jpeg_workspace_bytes(g, n_images);
    if (!d_workspace || workspace_bytes < need || ((uintptr_t)d_workspace & 15)) return FFHIP_EINVAL;
    const long long mcus = (long long)g->mcu_cols * g->mcu_rows;
    int16_t *sy = (int16_t *)d_workspace;
    int16_t *su = sy + mcus * g->h * g->v * 64 * n_images;
    int16_t *sv = su + mcus * 64 * n_images;
    for (int comp = 0; comp < g->ncomp; comp++) {
        IdctPlanes ip;
        ip.coef = comp == 0 ? d_coef_y : (comp == 1 ? d_coef_u : d_coef_v);
        ip.samples = comp == 0 ? sy : (comp == 1 ? su : sv);
        ip.quant = d_quant; ip.quant_stride = quant_stride;
        ip.blocks_per_image = comp == 0 ? mcus * g->h * g->v : mcus;
        ip.total_blocks = ip.blocks_per_image * n_images;
        ip.qt = g->qt_id[comp];
        long long rounds = (ip.total_blocks + 7) / 8;
        int grid = grid_for((rounds + WAVES_PER_WG - 1) / WAVES_PER_WG);
        hipLaunchKernelGGL(k_jpeg_idct_planes, dim3(grid), dim3(WG_THREADS), 0, st, ip);
        FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    }
    ColorGeneric cg;
    cg.sy = sy; cg.su = g->ncomp == 3 ? su : nullptr; cg.sv = g->ncomp == 3 ? sv : nullptr;
    cg.bgra = d_bgra; cg.pitch = pitch; cg.image_stride = image_stride;
    cg.mcu_cols = g->mcu_cols; cg.mcu_rows = g->mcu_rows; cg.h = g->h; cg.v = g->v; cg.n_images = n_images;
    long long items = (long long)width / 4 * height * n_images;
    int grid = grid_for((items + 255) / 256);
    hipLaunchKernelGGL(k_jpeg_color_generic, dim3(grid), dim3(256), 0, st, cg);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}

extern "C" int ffhip_jpeg_recon_batch(const ffhip_jpeg_geom *g, int n_images, const int16_t *d_coef_y,
                                      const int16_t *d_coef_u, const int16_t *d_coef_v,
                                      const uint16_t *d_quant, int64_t quant_stride, uint8_t *d_bgra,
                                      int64_t pitch, int64_t image_stride, void *d_workspace,
                                      size_t workspace_bytes, void *stream)
{
    return jpeg_recon_batch_impl(g, n_images, d_coef_y, d_coef_u, d_coef_v, d_quant, quant_stride, d_bgra, pitch, image_stride, d_workspace, workspace_bytes, stream, false);
}
/* Calibration (bench.py's roofline.pattern_GBps): the fused kernel's loads and stores on the caller's buffers with NO arithmetic in between -- same
 * grid, same workgroup-to-XCD mapping, same addresses and masks; d_bgra receives meaningless bytes.  Every layout a fused kernel takes (4:2:0 and the
 * strip layouts); FFHIP_EINVAL for the two-pass geometries. */
extern "C" int ffhip_jpeg_pattern_calibrate(const ffhip_jpeg_geom *g, int n_images, const int16_t *d_coef_y, const int16_t *d_coef_u, const int16_t *d_coef_v,
                                            const uint16_t *d_quant, int64_t quant_stride, uint8_t *d_bgra, int64_t pitch, int64_t image_stride, void *stream)
{
    return jpeg_recon_batch_impl(g, n_images, d_coef_y, d_coef_u, d_coef_v, d_quant, quant_stride, d_bgra, pitch, image_stride, nullptr, 0, stream, true);
}

#define SCRATCH_JPEG_HOST 6
extern "C" int ffhip_jpeg_recon_batch_host(const ffhip_jpeg_geom *g, int n_images, const int16_t *coef_y,
                                           const int16_t *coef_u, const int16_t *coef_v,
                                           const uint16_t *quant, int64_t quant_stride, uint8_t *bgra,
                                           int64_t pitch, int64_t image_stride)
{
    if (!geom_ok(g) || n_images < 0) return FFHIP_EINVAL;
    if (n_images == 0) return FFHIP_OK;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    /* one shared staging allocation: concurrent callers take turns for the whole call (copy in, launch, copy out) */
    static std::mutex host_call_mu;
    std::lock_guard<std::mutex> host_call_lock(host_call_mu);
    const size_t mcus = (size_t)g->mcu_cols * g->mcu_rows;
    const size_t ybytes = mcus * g->h * g->v * 128 * n_images, cbytes = mcus * 128 * n_images;
    const size_t qbytes = (quant_stride ? (size_t)quant_stride * (n_images - 1) + 256 : 256) * 2;
    const int64_t height = (int64_t)g->mcu_rows * 8 * g->v, row_bytes = 32LL * g->h * g->mcu_cols;
    if (pitch < row_bytes || (n_images > 1 && image_stride < pitch * height)) return FFHIP_EINVAL;
    /* the device buffer is the library's: it has the pitch ffhip_bgra_layout recommends (a slow output placement costs the fused kernel up to
     * 8 % at the reference's pitch), and the pixels reach the caller's layout by a 2-D copy */
    int64_t dpitch = 0, dstride = 0;
    if (ffhip_bgra_layout(g, &dpitch, &dstride) != FFHIP_OK) return FFHIP_EINVAL;
    const size_t obytes = (size_t)dstride * (size_t)n_images;
    const size_t wbytes = ffhip_jpeg_workspace_bytes(g, n_images);
    /* the device buffers are library scratch kept between calls (one allocation, grown on demand, released by
     * ffhip_shutdown): this is the per-picture entry a patched format/jpg.c calls, and allocating and freeing ~60 MB of
     * device memory around every picture cost more than the reconstruction itself.  Calls are serialised by the
     * mutex above, like the reference's single-threaded decode loop. */
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_u = up(ybytes), o_v = o_u + up(g->ncomp == 3 ? cbytes : 0), o_q = o_v + up(g->ncomp == 3 ? cbytes : 0);
    const size_t o_out = o_q + up(qbytes), o_ws = o_out + up(obytes), total = o_ws + up(wbytes);
    uint8_t *base = (uint8_t *)ffhip_scratch(SCRATCH_JPEG_HOST, nullptr, total / 4 + 64);
    if (!base) return FFHIP_ENOMEM;
    void *dy = base, *du = g->ncomp == 3 ? base + o_u : nullptr, *dv = g->ncomp == 3 ? base + o_v : nullptr, *dq = base + o_q, *dout = base + o_out;
    void *dws = wbytes ? base + o_ws : nullptr;
    FFHIP_CHECK(hipMemcpy(dy, coef_y, ybytes, hipMemcpyHostToDevice), FFHIP_EIO);
    if (g->ncomp == 3) {
        FFHIP_CHECK(hipMemcpy(du, coef_u, cbytes, hipMemcpyHostToDevice), FFHIP_EIO);
        FFHIP_CHECK(hipMemcpy(dv, coef_v, cbytes, hipMemcpyHostToDevice), FFHIP_EIO);
    }
    FFHIP_CHECK(hipMemcpy(dq, quant, qbytes, hipMemcpyHostToDevice), FFHIP_EIO);
    const int rc = ffhip_jpeg_recon_batch(g, n_images, (const int16_t *)dy, (const int16_t *)du, (const int16_t *)dv,
                                          (const uint16_t *)dq, quant_stride, (uint8_t *)dout, dpitch, dstride, dws, wbytes, nullptr);
    if (rc) return rc;
    if (n_images == 1 || image_stride == pitch * height) /* the caller's pictures follow each other row after row: one copy */
        FFHIP_CHECK(hipMemcpy2D(bgra, (size_t)pitch, dout, (size_t)dpitch, (size_t)row_bytes, (size_t)(height * n_images), hipMemcpyDeviceToHost), FFHIP_EIO);
    else
        for (int i = 0; i < n_images; i++)
            FFHIP_CHECK(hipMemcpy2D(bgra + (int64_t)i * image_stride, (size_t)pitch, (const uint8_t *)dout + (int64_t)i * dstride, (size_t)dpitch, (size_t)row_bytes,
                                    (size_t)height, hipMemcpyDeviceToHost), FFHIP_EIO);
    return FFHIP_OK;
}
