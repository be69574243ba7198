/*
 * ffhip_hevc.hip -- HEVC residual stage for batches of transform units of one size:
 * scaling (dequantisation) and the 2-D inverse transform, bit-exact with
 *   scale_transform_coefficients   coding/hevc.c:3743-3816
 *   transformation                 coding/hevc.c:3819-3885 (matrix :3826-3859)
 *   transform_scaled_coeffients    coding/hevc.c:3888-3956
 *   idct_4x4_hevc (luma intra 4x4) utils/idct.c:9-55, with its `+ (shift-1)` rounding
 *   transform-skip / bypass glue   coding/hevc.c:4209-4236
 *
 * HBM-bound: 2 B of levels in + 2 B of residual out per sample.  Levels come in and residuals go out as
 * linear 16-byte-per-lane copies through the wave's LDS tile (a row per lane would make every load a
 * 64-lane gather with a 2N-byte stride: 1.5 TB/s at N = 32).
 *
 * Default kernels:
 *   32x32, 16x16, 8x8  k_hevc_residual{32,16,8}_mfma: both 1-D passes as exact int8 x int8 -> int32 MFMA
 *                      products of the H.265 matrix (|entries| <= 90) with the int16 data split into a
 *                      high and a low byte; the first pass's accumulator tile is the second pass's operand
 *   4x4                k_hevc_residual4: one TU per lane in registers (DCT or DST-VII picked per lane)
 * Earlier form, kept for A/B runs (FFHIP_HEVC_RES32/16/8=dot, FFHIP_HEVC_RES4=rows) and tested alike:
 *   k_hevc_residual<N>: a wave holds 64 rows = 64/N TUs, one lane owns one column (then one row); the
 *   transposes go through LDS with ds_read_b64_tr_b16, whose row order is the bit-reversal-like order of
 *   the partial butterflies so each lane receives ready-made (x_a, x_b) int16 pairs for v_dot2_i32_i16;
 *   the N-point transform is the recursive even/odd decomposition of the H.265 matrix (N/2-point on the
 *   even inputs plus an N/2 x N/2 odd part), all accumulated mod 2^32 like the reference's int.
 */
#include "ffhip_internal.h"
#include <cstdlib>
#include <cstring>

#define PK16(lo, hi) ((u32)(uint16_t)(int16_t)(lo) | ((u32)(uint16_t)(int16_t)(hi) << 16))

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int dot2(u32 a, u32 b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
/* Heads of a dot-product chain.  The compiler's choice, the accumulate-in-place v_dot2c (it takes the
 * coefficient pair as a literal), costs a v_mov to seed every chain; the three-operand form with the
 * coefficients in an SGPR starts from the inline 0 or from a register that holds the rounding term. */
__device__ __forceinline__ int dot2_head0(u32 a, u32 coef)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(coef));
    return r;
}
__device__ __forceinline__ int dot2_head(u32 a, u32 coef, int seed)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(coef), "v"(seed));
    return r;
}

/* H.265 8.6.4.2: transMatrix[j][i] = c((2i+1) j mod 128), c(n) tabulated for n = 0..32 and
 * extended by c(64-n) = -c(n), c(64+n) = -c(n) */
__host__ __device__ constexpr int kcos(int n)
{
    constexpr int t[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                           61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
    return t[n];
}
__host__ __device__ constexpr int dct_coef(int j, int i)
{
    if (j == 0) return 64;
    const int n = ((2 * i + 1) * j) & 127;
    return n <= 32 ? kcos(n) : (n <= 64 ? -kcos(64 - n) : (n <= 96 ? -kcos(n - 64) : kcos(128 - n)));
}

/* Input row order of the recursive butterflies: order<N> = order<N/2> doubled, then the odd
 * indices.  order<4> = 0 2 1 3; order<8> = 0 4 2 6 1 3 5 7; ... */
template <int N>
__host__ __device__ constexpr int in_order(int k)
{
    if constexpr (N == 2) return k;
    else return k < N / 2 ? 2 * in_order<N / 2>(k) : 2 * (k - N / 2) + 1;
}

/* N-point inverse DCT of packed pairs in[k] = (x[order(2k)], x[order(2k+1)]); the matrix
 * row stride 32/N of hevc.c:3881 is folded into dct_coef's j argument. */
template <int N>
struct InvDct {
    static __device__ __forceinline__ void run(const u32 *in, int *out, int rnd)
    {
        int e[N / 2];
        InvDct<N / 2>::run(in, e, rnd);
        constexpr int s = 32 / N;
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            int o = dot2_head0(in[N / 4], PK16(dct_coef(1 * s, i), dct_coef(3 * s, i)));
#pragma unroll
            for (int k = 1; k < N / 4; k++) /* odd inputs x[4k+1], x[4k+3] */
                o = dot2(in[N / 4 + k], PK16(dct_coef((4 * k + 1) * s, i), dct_coef((4 * k + 3) * s, i)), o);
            out[i] = e[i] + o;
            out[N - 1 - i] = e[i] - o;
        }
    }
};
template <>
struct InvDct<4> {
    static __device__ __forceinline__ void run(const u32 *in, int *out, int rnd)
    {
        /* in[0] = (x0, x2), in[1] = (x1, x3); matrix rows 0, 8, 16, 24 */
        const int e0 = dot2_head(in[0], PK16(64, 64), rnd), e1 = dot2_head(in[0], PK16(64, -64), rnd);
        const int o0 = dot2_head0(in[1], PK16(83, 36)), o1 = dot2_head0(in[1], PK16(36, -83));
        out[0] = e0 + o0; out[1] = e1 + o1; out[2] = e1 - o1; out[3] = e0 - o0;
    }
};

/* DST-VII 4-point (idct.c:11-16): in[0] = (x0, x2), in[1] = (x1, x3) */
__device__ __forceinline__ void inv_dst4(const u32 *in, int *out, int rnd)
{
    out[0] = dot2(in[0], PK16(29, 84), dot2(in[1], PK16(74, 55), rnd));
    out[1] = dot2(in[0], PK16(55, -29), dot2(in[1], PK16(74, -84), rnd));
    out[2] = dot2(in[0], PK16(74, -74), dot2(in[1], PK16(0, 74), rnd));
    out[3] = dot2(in[0], PK16(84, 55), dot2(in[1], PK16(-74, -29), rnd));
}

struct HevcResArgs {
    const int16_t *level; /* [n_tu][N*N], row-major x + y*N                               */
    const uint8_t *tuinfo; /* [n_tu][4]: qP, flags, matrixId, 0                             */
    const uint8_t *scaling; /* [6][N*N] ScalingFactor (row-major) or NULL for flat m = 16  */
    int16_t *res;          /* [n_tu][N*N]                                                   */
    long long n_tu;
    int bitdepth, epp;
    int iters; /* batches per wave */
};

#define TU_DST 1u    /* trType 1: luma intra 4x4 -> idct_4x4_hevc                     */
#define TU_TSKIP 2u  /* transform_skip_flag: r = d << (5 + log2 N)                    */
#define TU_BYPASS 4u /* cu_transquant_bypass_flag: r = level                          */
#define TU_ROTATE 8u /* rotateCoeffs (4x4 intra with transform_skip_rotation_enabled) */

__device__ __forceinline__ int clip3i(int lo, int hi, int v) /* lo <= hi: the median */
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}
/* three-operand form (the compiler picks the accumulate-in-place v_dot2c and pays a v_mov to seed it) */
__device__ __forceinline__ int dot2_seed(u32 a, u32 b, int c)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* clip to [cmin, cmax] and keep the low 16 bits of both, packed; with the 16-bit coefficient range
 * (NARROW) that is one saturating v_cvt_pk_i16_i32 */
template <bool NARROW>
__device__ __forceinline__ u32 clip_pack(int v0, int v1, int cmin, int cmax)
{
    if (NARROW) return __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pk_i16(v0, v1));
    return __builtin_amdgcn_perm((u32)clip3i(cmin, cmax, v1), (u32)clip3i(cmin, cmax, v0), 0x05040100u);
}

/* hevc.c:3786-3805 on CH consecutive samples of one TU: d = clip3((level * m * levelScale[qP % 6] << (qP / 6)) + rnd >> bdShift).
 * |level| < 2^15 and m * levelScale <= 255 * 72, so the product is an exact full-rate v_mul_i32_i24; the shift and the
 * rounding add are one v_lshl_add_u32 (mod 2^32 like the reference's int) */
template <int CH, bool NARROW, bool FLAT>
__device__ __forceinline__ void scale_chunk_f(const u32 *raw, u32 *outd, int ls, u32 sh, const uint8_t *mrow, int bd_shift, int cmin,
                                              int cmax)
{
    u32 rnd = 1u << (bd_shift - 1);
    asm("" : "+v"(rnd)); /* in a VGPR, so that shift + add is one v_lshl_add_u32 (one SGPR operand per instruction) */
    u32 mraw[CH / 4] = {};
    if (!FLAT) {
#pragma unroll
        for (int i = 0; i < CH / 4; i++) mraw[i] = *(const u32 *)(mrow + 4 * i);
    }
    int dv[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        const int lv = (i & 1) ? (int)raw[i >> 1] >> 16 : (int)(short)(raw[i >> 1] & 0xffffu);
        int f = FLAT ? 16 * ls : __mul24((int)((mraw[i >> 2] >> (8 * (i & 3))) & 0xffu), ls);
        if (!FLAT) asm("" : "+v"(f)); /* keep (m * ls) * level: the other association needs a 32-bit multiply */
        dv[i] = (int)(((u32)__mul24(lv, f) << sh) + rnd) >> bd_shift;
    }
#pragma unroll
    for (int i = 0; i < CH / 2; i++) outd[i] = clip_pack<NARROW>(dv[2 * i], dv[2 * i + 1], cmin, cmax);
}
template <int CH, bool NARROW>
__device__ __forceinline__ void scale_chunk(const u32 *raw, u32 *outd, int qP, bool flat, const uint8_t *mrow, int bd_shift,
                                            int cmin, int cmax)
{
    const int ls = qP % 6 == 0 ? 40 : (qP % 6 == 1 ? 45 : (qP % 6 == 2 ? 51 : (qP % 6 == 3 ? 57 : (qP % 6 == 4 ? 64 : 72))));
    if (!flat) scale_chunk_f<CH, NARROW, false>(raw, outd, ls, qP / 6, mrow, bd_shift, cmin, cmax);
    else scale_chunk_f<CH, NARROW, true>(raw, outd, ls, qP / 6, mrow, bd_shift, cmin, cmax);
}

/* What one lane fetches for one batch of 64 rows: NCH chunks of CH consecutive samples + the chunk's TU descriptor */
template <int N>
struct ResFetch {
    static constexpr int CH = N >= 8 ? 8 : 4, NCH = N / CH;
    u32 raw[NCH][CH / 2];
    u32 inf[NCH];
    u32 tinf; /* the descriptor of TU tu0 + lane / N: the lane's TU in the two passes */
    __device__ __forceinline__ void issue(const HevcResArgs &a, long long tu0, u32 lane)
    {
        {
            long long tu = tu0 + lane / N;
            if (tu >= a.n_tu) tu = a.n_tu - 1;
            tinf = *(const u32 *)(a.tuinfo + tu * 4);
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const u32 s0 = (64u * c + lane) * CH, tul = s0 / (N * N), pos = s0 % (N * N);
            long long tuc = tu0 + tul;
            if (tuc >= a.n_tu) tuc = a.n_tu - 1;
            inf[c] = *(const u32 *)(a.tuinfo + tuc * 4);
            if (CH == 8) {
                const u32x4 v = __builtin_nontemporal_load((const u32x4 *)(a.level + tuc * (N * N) + pos));
                raw[c][0] = v[0]; raw[c][1] = v[1]; raw[c][CH / 2 - 2] = v[2]; raw[c][CH / 2 - 1] = v[3];
            } else {
                const u32x2 v = __builtin_nontemporal_load((const u32x2 *)(a.level + tuc * (N * N) + pos));
                raw[c][0] = v[0]; raw[c][1] = v[1];
            }
        }
    }
};

template <int N, bool NARROW>
__global__ __launch_bounds__(256) void k_hevc_residual(HevcResArgs a)
{
    constexpr int W = N / 2;          /* dwords per row                       */
    constexpr int TPW = 64 / N;       /* TUs per wave and batch               */
    constexpr int LOG2N = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
    constexpr int CH = ResFetch<N>::CH, NCH = ResFetch<N>::NCH; /* samples per lane per pass, passes */
    __shared__ __attribute__((aligned(16))) char lds_all[4 * 64 * N * 2];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *tile = lds_all + wave * (64 * N * 2);
    const u32 tu_l = lane / N, idx = lane % N; /* TU within the batch, row (load/store) or column/row (passes) */
    const int range = a.epp ? (a.bitdepth + 6 > 15 ? a.bitdepth + 6 : 15) : 15;
    const int cmin = -(1 << range), cmax = (1 << range) - 1;
    u32 *trow = (u32 *)(tile + (tu_l * N + idx) * (N * 2)); /* this lane's row of its TU */
    /* transposing reads: 16-lane group gq, lane 4q+p supplies row order[4k+q], 4 columns */
    const u32 t16 = lane & 15, q = t16 >> 2, p = t16 & 3, g16 = lane >> 4;
    u32 col_tu, col_x0;
    if (N >= 16) { col_tu = g16 / (N / 16 > 0 ? N / 16 : 1); col_x0 = 16 * (g16 % (N / 16 > 0 ? N / 16 : 1)) + 4 * p; }
    else if (N == 8) { col_tu = 2 * g16 + (p >> 1); col_x0 = 4 * (p & 1); }
    else { col_tu = 4 * g16 + p; col_x0 = 0; }
    const char *tr_base = tile + col_tu * (N * N * 2) + col_x0 * 2;

    /* A wave works through a.iters batches of 64 rows, the next batch's levels in flight while this one
     * is transformed (one batch per wave left the kernel bound by wave launches at the small sizes). */
    const long long wave_tu0 = ((long long)blockIdx.x * 4 + wave) * TPW * a.iters;
    ResFetch<N> nxt;
    if (wave_tu0 < a.n_tu) nxt.issue(a, wave_tu0, lane);
    for (int it = 0; it < a.iters; it++) {
    const long long tu0 = wave_tu0 + (long long)it * TPW;
    if (tu0 >= a.n_tu) break;
    const ResFetch<N> cur = nxt;
    if (it + 1 < a.iters && tu0 + TPW < a.n_tu) nxt.issue(a, tu0 + TPW, lane);
    long long tu = tu0 + tu_l;
    const bool live = tu < a.n_tu;
    if (!live) tu = a.n_tu - 1; /* keep EXEC full for the transposing reads; stores are masked */
    const u32 flags = (cur.tinf >> 8) & 0xff;

    /* ---- levels -> scaled coefficients d (hevc.c:3786-3805) -> LDS, as a LINEAR copy: lane L of pass c
     * takes samples (64c + L)*CH .. +CH of the batch's 64 rows, so every load is 16 (8) contiguous bytes
     * per lane and the tile in LDS is simply row-major [tu][row][col] ---- */
    bool chunk_store[NCH];              /* does this chunk's TU take the transform path and exist? */
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u32 s0 = (64u * c + lane) * CH, tul = s0 / (N * N), pos = s0 % (N * N);
        const u32 inf = cur.inf[c];
        const int qP = inf & 0xff;
        const u32 fl = (inf >> 8) & 0xff, mid = (inf >> 16) & 0xff;
        chunk_store[c] = tu0 + tul < a.n_tu && !(fl & (TU_BYPASS | TU_TSKIP));
        u32 outd[CH / 2];
        if (fl & TU_BYPASS) {
#pragma unroll
            for (int i = 0; i < CH / 2; i++) outd[i] = cur.raw[c][i];
        } else {
            const bool flat = a.scaling == nullptr || ((fl & TU_TSKIP) && N > 4);
            scale_chunk<CH, NARROW>(cur.raw[c], outd, qP, flat, a.scaling + mid * (N * N) + pos, a.bitdepth + LOG2N + 10 - range,
                                    cmin, cmax);
        }
        if (CH == 8) {
            u32x4 w;
            w[0] = outd[0]; w[1] = outd[1]; w[2] = outd[CH / 2 - 2]; w[3] = outd[CH / 2 - 1];
            *(u32x4 *)(tile + s0 * 2) = w;
        } else {
            u32x2 w;
            w[0] = outd[0]; w[1] = outd[1];
            *(u32x2 *)(tile + s0 * 2) = w;
        }
    }
    if (__builtin_amdgcn_ballot_w64((flags & (TU_BYPASS | TU_TSKIP)) != 0)) {
        /* no transform for some TU of this wave: r = level, or d << tsShift; optional 180-degree rotation
         * (hevc.c:4209-4236).  Row-per-lane, rare: the row comes back from LDS */
        if (flags & (TU_BYPASS | TU_TSKIP)) {
            const int ts = (flags & TU_TSKIP) ? 5 + LOG2N : 0;
            int d[N];
#pragma unroll
            for (int i = 0; i < W; i++) { const u32 v = trow[i]; d[2 * i] = (int)(short)(v & 0xffffu); d[2 * i + 1] = (int)v >> 16; }
            u32 outp[W];
#pragma unroll
            for (int i = 0; i < W; i++) {
                const int lo = (flags & TU_ROTATE) ? d[N - 1 - 2 * i] : d[2 * i];
                const int hi = (flags & TU_ROTATE) ? d[N - 2 - 2 * i] : d[2 * i + 1];
                outp[i] = ((u32)(lo << ts) & 0xffffu) | ((u32)(hi << ts) << 16);
            }
            if (live) {
                const int orow = (flags & TU_ROTATE) ? N - 1 - (int)idx : (int)idx;
                u32 *dst = (u32 *)(a.res + tu * (N * N) + orow * N);
#pragma unroll
                for (int i = 0; i < W; i++) __builtin_nontemporal_store(outp[i], dst + i);
            }
        }
    }

    u32 pairs[W];
    int e[N];
    /* ---- first stage: columns (hevc.c:3931-3939) ---- */
#pragma unroll
    for (int k = 0; k < N / 4; k++) {
        const int r0 = in_order<N>(4 * k), r1 = in_order<N>(4 * k + 1), r2 = in_order<N>(4 * k + 2), r3 = in_order<N>(4 * k + 3);
        const int rq = q == 0 ? r0 : (q == 1 ? r1 : (q == 2 ? r2 : r3));
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tr_base + rq * (N * 2)));
        const u32x2 vv = __builtin_bit_cast(u32x2, v);
        pairs[2 * k] = vv[0];
        pairs[2 * k + 1] = vv[1];
    }
    const bool dst = (flags & TU_DST) != 0;
    if (N == 4) {
        int ed[4];
        InvDct<N>::run(pairs, e, 64);
        inv_dst4(pairs, ed, 6); /* idct.c:31: + (shift - 1) with shift 7 */
#pragma unroll
        for (int i = 0; i < 4; i++) e[i] = dst ? ed[i] : e[i];
    } else {
        InvDct<N>::run(pairs, e, 64);
    }
    /* g = clip3(coeffMin, coeffMax, (e + 64) >> 7), int16; written as row `idx` of the
     * transposed tile so the second transposing read hands each lane one row of g */
#pragma unroll
    for (int i = 0; i < W; i++) {
        trow[i] = clip_pack<NARROW>(e[2 * i] >> 7, e[2 * i + 1] >> 7, cmin, cmax);
    }
    /* ---- second stage: rows (hevc.c:3943-3953) ---- */
#pragma unroll
    for (int k = 0; k < N / 4; k++) {
        const int r0 = in_order<N>(4 * k), r1 = in_order<N>(4 * k + 1), r2 = in_order<N>(4 * k + 2), r3 = in_order<N>(4 * k + 3);
        const int rq = q == 0 ? r0 : (q == 1 ? r1 : (q == 2 ? r2 : r3));
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tr_base + rq * (N * 2)));
        const u32x2 vv = __builtin_bit_cast(u32x2, v);
        pairs[2 * k] = vv[0];
        pairs[2 * k + 1] = vv[1];
    }
    int sh2 = 20 - a.bitdepth;
    if (a.epp && sh2 < 11) sh2 = 11;
    if (sh2 < 0) sh2 = 0;
    int r[N];
    if (N == 4) {
        int rd[4];
        InvDct<N>::run(pairs, r, sh2 > 0 ? 1 << (sh2 - 1) : 0);
        inv_dst4(pairs, rd, sh2 - 1);
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = dst ? clip3i(cmin, cmax, rd[i] >> sh2) : r[i] >> sh2; /* DST clips both stages */
    } else {
        InvDct<N>::run(pairs, r, sh2 > 0 ? 1 << (sh2 - 1) : 0);
#pragma unroll
        for (int i = 0; i < N; i++) r[i] >>= sh2;
    }
    /* ---- rows of r -> LDS -> global as the same linear copy (the stage-2 reads were issued before these writes) ---- */
#pragma unroll
    for (int i = 0; i < W; i++) trow[i] = __builtin_amdgcn_perm((u32)r[2 * i + 1], (u32)r[2 * i], 0x05040100u);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u32 s0 = (64u * c + lane) * CH;
        if (chunk_store[c]) {
            int16_t *dstp = a.res + tu0 * (N * N) + s0;
            if (CH == 8) __builtin_nontemporal_store(*(const u32x4 *)(tile + s0 * 2), (u32x4 *)dstp);
            else __builtin_nontemporal_store(*(const u32x2 *)(tile + s0 * 2), (u32x2 *)dstp);
        }
    }
    } /* batches */
}

/* The six ScalingFactor lists of a TU size (6 N^2 bytes) staged in LDS by the workgroup, so that the per-sample m
 * is an LDS read behind the one barrier of the kernel instead of a global load that depends on the TU descriptor
 * and is consumed at once.  LISTED kernels only: the flat ones carry neither the LDS nor the barrier. */
template <int NN>
__device__ __forceinline__ void stage_scaling_lists(uint8_t *sl, const uint8_t *scaling)
{
    for (int i = threadIdx.x; i < 6 * NN * NN / 4; i += 256) ((u32 *)sl)[i] = ((const u32 *)scaling)[i];
    __syncthreads();
}

/* ------------------------------------------------------------------------------------------------
 * 32x32 TUs on the matrix cores.
 *
 * Both passes of hevc.c:3931-3953 are products with the 32x32 H.265 matrix M[k][i] = dct_coef(k, i):
 *   stage 1 (columns)  e[y][x] = sum_k M[k][y] d[k][x]      g = clip3((e + 64) >> 7)
 *   stage 2 (rows)     r[y][x'] = (sum_x g[y][x] M[x][x'] + rnd) >> sh2
 * computed transposed so that the first product's accumulator tile is the second product's operand with
 * no lane movement (C/D of a 32x32 MFMA: column on the lane, rows 8(i>>2) + 4h + (i&3) in register i,
 * h = lane >> 5):
 *   C1[x][y]  = sum_k A1[x][k] B1[k][y],   A1 = d^T (data),  B1 = M
 *   C2[x'][y] = sum_x A2[x'][x] B2[x][y],  A2 = M^T,         B2 = g^T = clip(C1)   (sums over C1's ROW index)
 * The sum index of a product may be permuted freely as long as both operands agree, so element j of
 * lane half h is k = 8(j>>2) + 4h + (j&3) everywhere: that is the order C1 delivers, B1 and A2 become
 * the SAME four registers (kTab.m), and one ds_read_b64_tr_b16 of the row-major LDS tile fetches four
 * consecutive j of A1.
 * int16 data v is fed as two int8 operands: v = 256 hi + lo' + 128 with hi = v >> 8, lo' = (v & 255) - 128
 * (byte pattern lo ^ 0x80), so sum M v = 256 (M.hi) + (M.lo') + 128 sum M -- two MFMAs per pass, the
 * constant (plus the rounding term) preloaded as the C operand of the low one.  Every partial sum is
 * below 2^27: exact.
 * One wave owns one TU at a time, so the per-TU flags are wave-uniform: transform-skip and bypass TUs
 * leave straight from the scaling registers.
 * ------------------------------------------------------------------------------------------------ */
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct Mfma32Tab {
    u32 m[64][4]; /* lane (h, r): byte j = M[8(j>>2) + 4h + (j&3)][r] */
    int s[32];    /* 128 * sum_k M[k][i]                                */
    int c2[2][16]; /* s at the rows register i of lane half h holds     */
};
constexpr Mfma32Tab make_mfma32_tab()
{
    Mfma32Tab t = {};
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) {
            const int k = 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
            t.m[l][j >> 2] |= (u32)(uint8_t)(int8_t)dct_coef(k, l & 31) << (8 * (j & 3));
        }
    for (int i = 0; i < 32; i++) {
        int sum = 0;
        for (int k = 0; k < 32; k++) sum += dct_coef(k, i);
        t.s[i] = 128 * sum;
    }
    for (int h = 0; h < 2; h++)
        for (int i = 0; i < 16; i++) t.c2[h][i] = t.s[8 * (i >> 2) + 4 * h + (i & 3)];
    return t;
}
__device__ const Mfma32Tab kTab = make_mfma32_tab();

/* int16 pairs (j0,j1), (j2,j3) -> the four high bytes, the four low bytes biased by -128 */
__device__ __forceinline__ void split_bytes(u32 d0, u32 d1, int &hi, int &lo)
{
    hi = (int)__builtin_amdgcn_perm(d1, d0, 0x07050301u);
    lo = (int)(__builtin_amdgcn_perm(d1, d0, 0x06040200u) ^ 0x80808080u);
}

template <bool NARROW, bool LISTED>
__global__ __launch_bounds__(256) void k_hevc_residual32_mfma(HevcResArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t sl[LISTED ? 6 * 32 * 32 : 16];
    if (LISTED) stage_scaling_lists<32>(sl, a.scaling);
    constexpr int N = 32, TPW = 2, OST = 72; /* TUs per wave; byte stride of the output tile's rows (bank spread) */
    __shared__ __attribute__((aligned(16))) char lds_in[4][TPW][N * N * 2];
    __shared__ __attribute__((aligned(16))) char lds_out[4][TPW][N * OST];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 h = lane >> 5, col = lane & 31;
    const int range = a.epp ? (a.bitdepth + 6 > 15 ? a.bitdepth + 6 : 15) : 15;
    const int cmin = -(1 << range), cmax = (1 << range) - 1;
    const int bd_shift = a.bitdepth + 5 + 10 - range;
    int sh2 = 20 - a.bitdepth;
    if (a.epp && sh2 < 11) sh2 = 11;
    if (sh2 < 0) sh2 = 0;
    const int rnd2 = sh2 > 0 ? 1 << (sh2 - 1) : 0;
    const v4i mreg = *(const v4i *)kTab.m[lane];
    v16i c1, c2;
    {
        const int s1 = kTab.s[col] + 64;
#pragma unroll
        for (int i = 0; i < 16; i++) { c1[i] = s1; c2[i] = kTab.c2[h][i] + rnd2; }
    }
    /* transposing read: lane 4q+p of a 16-lane group supplies row 8t + 4h + q, columns 16(g&1) + 4p .. +3 */
    const u32 t16 = lane & 15, q = t16 >> 2, p = t16 & 3, g16 = lane >> 4;
    const u32 tr_off = (4 * h + q) * (N * 2) + (16 * (g16 & 1) + 4 * p) * 2;

    /* a.iters batches of TPW TUs per wave, the next batch's levels in flight during this one's transform */
    const long long wave_tu0 = ((long long)blockIdx.x * 4 + wave) * TPW * a.iters;
    u32x4 nraw[TPW][2];
    u32 ninf[TPW];
    auto issue = [&](long long t0) {
#pragma unroll
        for (int u = 0; u < TPW; u++) {
            const long long tu = t0 + u < a.n_tu ? t0 + u : a.n_tu - 1;
            ninf[u] = *(const u32 *)(a.tuinfo + tu * 4);
#pragma unroll
            for (int c = 0; c < 2; c++)
                nraw[u][c] = __builtin_nontemporal_load((const u32x4 *)(a.level + tu * (N * N) + (64u * c + lane) * 8));
        }
    };
    if (wave_tu0 < a.n_tu) issue(wave_tu0);
    for (int it = 0; it < a.iters; it++) {
    const long long tu0 = wave_tu0 + (long long)it * TPW;
    if (tu0 >= a.n_tu) break;
    u32x4 craw[TPW][2];
    u32 cinf[TPW];
#pragma unroll
    for (int u = 0; u < TPW; u++) { craw[u][0] = nraw[u][0]; craw[u][1] = nraw[u][1]; cinf[u] = ninf[u]; }
    if (it + 1 < a.iters && tu0 + TPW < a.n_tu) issue(tu0 + TPW);

    /* ---- levels (linear 16 B per lane), scaled; transform TUs go to the LDS tile, the others leave ---- */
    bool xform[TPW];
#pragma unroll
    for (int u = 0; u < TPW; u++) {
        const long long tu = tu0 + u;
        xform[u] = false;
        if (tu >= a.n_tu) continue;
        const u32 inf = (u32)__builtin_amdgcn_readfirstlane((int)cinf[u]);
        const int qP = inf & 0xff;
        const u32 fl = (inf >> 8) & 0xff, mid = (inf >> 16) & 0xff;
        xform[u] = !(fl & (TU_BYPASS | TU_TSKIP));
        const bool flat = !LISTED || (fl & TU_TSKIP);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const u32 pos = (64u * c + lane) * 8;
            const u32x4 v = craw[u][c];
            u32 raw[4] = {v[0], v[1], v[2], v[3]}, outd[4];
            if (fl & TU_BYPASS) {
#pragma unroll
                for (int i = 0; i < 4; i++) outd[i] = raw[i];
            } else {
                scale_chunk<8, NARROW>(raw, outd, qP, flat, sl + mid * (N * N) + pos, bd_shift, cmin, cmax);
            }
            if (xform[u]) {
                *(u32x4 *)(lds_in[wave][u] + pos * 2) = u32x4{outd[0], outd[1], outd[2], outd[3]};
            } else {
                /* hevc.c:4209-4236: r = level, or d << tsShift; optional 180-degree rotation = the chunk
                 * reversed at the mirrored position */
                const int ts = (fl & TU_TSKIP) ? 5 + 5 : 0;
                u32 o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int lo = (int)(short)(outd[i] & 0xffffu) << ts, hi = ((int)outd[i] >> 16) << ts;
                    o[i] = __builtin_amdgcn_perm((u32)hi, (u32)lo, 0x05040100u);
                }
                u32x4 w = {o[0], o[1], o[2], o[3]};
                u32 opos = pos;
                if (fl & TU_ROTATE) {
#pragma unroll
                    for (int i = 0; i < 4; i++) w[i] = __builtin_amdgcn_perm(o[3 - i], o[3 - i], 0x01000302u);
                    opos = N * N - 8 - pos;
                }
                __builtin_nontemporal_store(w, (u32x4 *)(a.res + tu * (N * N) + opos));
            }
        }
    }

#pragma unroll
    for (int u = 0; u < TPW; u++) {
        if (!xform[u]) continue;
        const char *tin = lds_in[wave][u];
        char *tout = lds_out[wave][u];
        v4i ah, al;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tin + tr_off + t * 8 * (N * 2)));
            const u32x2 vv = __builtin_bit_cast(u32x2, v);
            int hi, lo;
            split_bytes(vv[0], vv[1], hi, lo);
            ah[t] = hi;
            al[t] = lo;
        }
        const v16i zero = {};
        v16i eh = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, mreg, zero, 0, 0, 0);
        v16i el = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, mreg, c1, 0, 0, 0);
        v4i bh, bl;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            u32 g[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i0 = 4 * t + 2 * j;
                g[j] = clip_pack<NARROW>(((eh[i0] << 8) + el[i0]) >> 7, ((eh[i0 + 1] << 8) + el[i0 + 1]) >> 7, cmin, cmax);
            }
            int hi, lo;
            split_bytes(g[0], g[1], hi, lo);
            bh[t] = hi;
            bl[t] = lo;
        }
        eh = __builtin_amdgcn_mfma_i32_32x32x32_i8(mreg, bh, zero, 0, 0, 0);
        el = __builtin_amdgcn_mfma_i32_32x32x32_i8(mreg, bl, c2, 0, 0, 0);
        /* register i of this lane is r[y = col][x' = 8(i>>2) + 4h + (i&3)]: four 8-byte pieces of row y */
#pragma unroll
        for (int t = 0; t < 4; t++) {
            u32x2 w;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i0 = 4 * t + 2 * j;
                w[j] = __builtin_amdgcn_perm((u32)(((eh[i0 + 1] << 8) + el[i0 + 1]) >> sh2), (u32)(((eh[i0] << 8) + el[i0]) >> sh2),
                                             0x05040100u);
            }
            *(u32x2 *)(tout + col * OST + (8 * t + 4 * h) * 2) = w;
        }
        /* out as the same linear copy */
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const u32 pos = (64u * c + lane) * 8;
            const char *src = tout + (pos >> 5) * OST + (pos & 31) * 2;
            const u32x2 w0 = *(const u32x2 *)src, w1 = *(const u32x2 *)(src + 8);
            __builtin_nontemporal_store(u32x4{w0[0], w0[1], w1[0], w1[1]}, (u32x4 *)(a.res + (tu0 + u) * (N * N) + pos));
        }
    }
    } /* batches */
}

/* ------------------------------------------------------------------------------------------------
 * 4x4 TUs: one TU per lane, everything in registers (32 B in, 32 B out per lane, consecutive lanes ->
 * consecutive TUs, so the wave's traffic is one linear 2 KB stream each way and there is no LDS).
 * With 64 rows per wave (k_hevc_residual<4>) a lane had four samples per batch and ~60 instructions
 * of bookkeeping per sample.  Both 4-point transforms are the same eight dot products with the matrix
 * picked per lane: DCT (hevc.c:3826-3859 rows 0, 8, 16, 24) or DST-VII (idct.c:11-16).
 * ------------------------------------------------------------------------------------------------ */
template <bool NARROW>
__global__ __launch_bounds__(256) void k_hevc_residual4(HevcResArgs a)
{
    const int range = a.epp ? (a.bitdepth + 6 > 15 ? a.bitdepth + 6 : 15) : 15;
    const int cmin = -(1 << range), cmax = (1 << range) - 1;
    const int bd_shift = a.bitdepth + 2 + 10 - range;
    int sh2 = 20 - a.bitdepth;
    if (a.epp && sh2 < 11) sh2 = 11;
    if (sh2 < 0) sh2 = 0;
    {
        /* the wave's 64 TUs are one 2 KB run: two linear 1 KB loads (16 B per lane), then each lane takes its TU's
         * 32 bytes back from LDS -- a direct 32-byte-stride load uses half of every 64-byte segment per instruction */
        __shared__ __attribute__((aligned(16))) char lds4[4][64 * 32];
        const u32 lane = threadIdx.x & 63;
        char *tile = lds4[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
        const long long wtu0 = (long long)blockIdx.x * 256 + (threadIdx.x & ~63u);
        if (wtu0 >= a.n_tu) return;
        const long long n_samples = a.n_tu * 16;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const long long s0 = wtu0 * 16 + (64 * c + lane) * 8;
            u32x4 v = {};
            if (s0 < n_samples) v = __builtin_nontemporal_load((const u32x4 *)(a.level + s0));
            *(u32x4 *)(tile + (64 * c + lane) * 16) = v;
        }
        long long tu = wtu0 + lane;
        const bool live = tu < a.n_tu;
        if (!live) tu = a.n_tu - 1;
        const u32 inf = *(const u32 *)(a.tuinfo + tu * 4);
        const u32x4 v0 = *(const u32x4 *)(tile + lane * 32), v1 = *(const u32x4 *)(tile + lane * 32 + 16);
        const int qP = inf & 0xff;
        const u32 fl = (inf >> 8) & 0xff, mid = (inf >> 16) & 0xff;
        const u32 raw[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        /* scaling (hevc.c:3786-3805), kept as ints d[x + 4y] */
        const int ls = qP % 6 == 0 ? 40 : (qP % 6 == 1 ? 45 : (qP % 6 == 2 ? 51 : (qP % 6 == 3 ? 57 : (qP % 6 == 4 ? 64 : 72))));
        const u32 sh = qP / 6;
        u32 rnd = 1u << (bd_shift - 1);
        asm("" : "+v"(rnd));
        int d[16];
        if (a.scaling) {
            const u32x4 mv = *(const u32x4 *)(a.scaling + mid * 16);
            const u32 m[4] = {mv[0], mv[1], mv[2], mv[3]};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int lv = (i & 1) ? (int)raw[i >> 1] >> 16 : (int)(short)(raw[i >> 1] & 0xffffu);
                int f = __mul24((int)((m[i >> 2] >> (8 * (i & 3))) & 0xffu), ls);
                asm("" : "+v"(f));
                d[i] = (int)(((u32)__mul24(lv, f) << sh) + rnd) >> bd_shift;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int lv = (i & 1) ? (int)raw[i >> 1] >> 16 : (int)(short)(raw[i >> 1] & 0xffffu);
                d[i] = (int)(((u32)__mul24(lv, 16 * ls) << sh) + rnd) >> bd_shift;
            }
        }
        const bool dst = (fl & TU_DST) != 0;
        u32 c0[4], c1[4]; /* c0[i] = (M[0][i], M[2][i]), c1[i] = (M[1][i], M[3][i]) */
        c0[0] = dst ? PK16(29, 84) : PK16(64, 64);   c1[0] = dst ? PK16(74, 55) : PK16(83, 36);
        c0[1] = dst ? PK16(55, -29) : PK16(64, -64); c1[1] = dst ? PK16(74, -84) : PK16(36, -83);
        c0[2] = dst ? PK16(74, -74) : PK16(64, -64); c1[2] = dst ? PK16(0, 74) : PK16(-36, 83);
        c0[3] = dst ? PK16(84, 55) : PK16(64, 64);   c1[3] = dst ? PK16(-74, -29) : PK16(-83, -36);
        const int rnd1 = dst ? 6 : 64;                                  /* idct.c:31: + (shift - 1), shift 7 */
        const int rnd2 = dst ? sh2 - 1 : (sh2 > 0 ? 1 << (sh2 - 1) : 0);
        const int lo2 = dst ? cmin : (int)0x80000000, hi2 = dst ? cmax : 0x7fffffff; /* DST clips both stages */
        /* first stage, columns: e[c][y] */
        int e[4][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const u32 in0 = clip_pack<NARROW>(d[c], d[8 + c], cmin, cmax), in1 = clip_pack<NARROW>(d[4 + c], d[12 + c], cmin, cmax);
#pragma unroll
            for (int i = 0; i < 4; i++) e[c][i] = dot2(in0, c0[i], dot2_seed(in1, c1[i], rnd1)) >> 7;
        }
        /* second stage, rows */
        u32 out[8];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const u32 in0 = clip_pack<NARROW>(e[0][y], e[2][y], cmin, cmax), in1 = clip_pack<NARROW>(e[1][y], e[3][y], cmin, cmax);
            int r[4];
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = clip3i(lo2, hi2, dot2(in0, c0[i], dot2_seed(in1, c1[i], rnd2)) >> sh2);
            out[2 * y] = __builtin_amdgcn_perm((u32)r[1], (u32)r[0], 0x05040100u);
            out[2 * y + 1] = __builtin_amdgcn_perm((u32)r[3], (u32)r[2], 0x05040100u);
        }
        if (fl & (TU_BYPASS | TU_TSKIP)) {
            /* hevc.c:4209-4236: r = level, or d << tsShift (7 at this size); optional 180-degree rotation */
            u32 o[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const u32 dp = clip_pack<NARROW>(d[2 * i], d[2 * i + 1], cmin, cmax);
                const int lo = (int)(short)(dp & 0xffffu) << 7, hi = ((int)dp >> 16) << 7;
                o[i] = (fl & TU_BYPASS) ? raw[i] : __builtin_amdgcn_perm((u32)hi, (u32)lo, 0x05040100u);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) out[i] = (fl & TU_ROTATE) ? __builtin_amdgcn_perm(o[7 - i], o[7 - i], 0x01000302u) : o[i];
        }
        *(u32x4 *)(tile + lane * 32) = u32x4{out[0], out[1], out[2], out[3]};
        *(u32x4 *)(tile + lane * 32 + 16) = u32x4{out[4], out[5], out[6], out[7]};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const long long s0 = wtu0 * 16 + (64 * c + lane) * 8;
            const u32x4 v = *(const u32x4 *)(tile + (64 * c + lane) * 16);
            if (s0 < n_samples) __builtin_nontemporal_store(v, (u32x4 *)(a.res + s0));
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * 16x16 TUs on the matrix cores: the same scheme with v_mfma_i32_16x16x32_i8.  C/D there is "column on
 * the lane, rows 4g + i in register i" (g = lane >> 4), so element j < 4 of lane group g is sum index
 * 4g + j in every operand and elements 4..7 (the upper half of the K = 32 the instruction offers) are
 * zero bytes in the data operand.  A wave takes 64 rows = four TUs per batch like k_hevc_residual<16>
 * (same linear copies, same prefetch), then runs the TUs one after the other through the four MFMAs.
 * ------------------------------------------------------------------------------------------------ */
struct Mfma16Tab {
    u32 m[64]; /* lane (g, r): byte j = M16[4g + j][r]   */
    int s[16]; /* 128 * sum_k M16[k][i]                   */
};
constexpr Mfma16Tab make_mfma16_tab()
{
    Mfma16Tab t = {};
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 4; j++) t.m[l] |= (u32)(uint8_t)(int8_t)dct_coef(2 * (4 * (l >> 4) + j), l & 15) << (8 * j);
    for (int i = 0; i < 16; i++) {
        int sum = 0;
        for (int k = 0; k < 16; k++) sum += dct_coef(2 * k, i);
        t.s[i] = 128 * sum;
    }
    return t;
}
__device__ const Mfma16Tab kTab16 = make_mfma16_tab();

template <bool NARROW, bool LISTED>
__global__ __launch_bounds__(256) void k_hevc_residual16_mfma(HevcResArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t sl[LISTED ? 6 * 16 * 16 : 16];
    if (LISTED) stage_scaling_lists<16>(sl, a.scaling);
    constexpr int N = 16, TPW = 4, OST = 40; /* TUs per batch; byte stride of the output tile's rows */
    constexpr int CH = 8, NCH = 2;
    __shared__ __attribute__((aligned(16))) char lds_in[4][TPW * N * N * 2];
    __shared__ __attribute__((aligned(16))) char lds_out[4][TPW * N * OST];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *tin = lds_in[wave], *tout = lds_out[wave];
    const u32 g = lane >> 4, col = lane & 15;
    const int range = a.epp ? (a.bitdepth + 6 > 15 ? a.bitdepth + 6 : 15) : 15;
    const int cmin = -(1 << range), cmax = (1 << range) - 1;
    const int bd_shift = a.bitdepth + 4 + 10 - range;
    int sh2 = 20 - a.bitdepth;
    if (a.epp && sh2 < 11) sh2 = 11;
    if (sh2 < 0) sh2 = 0;
    const int rnd2 = sh2 > 0 ? 1 << (sh2 - 1) : 0;
    const long mreg = (long)(unsigned long)kTab16.m[lane]; /* upper four bytes: the unused half of K */
    v4i c1, c2;
#pragma unroll
    for (int i = 0; i < 4; i++) { c1[i] = kTab16.s[col] + 64; c2[i] = kTab16.s[4 * g + i] + rnd2; }
    /* transposing read: lane 4q+p of 16-lane group g supplies row 4g + q, columns 4p .. 4p+3 */
    const u32 q = (lane & 15) >> 2, p = lane & 3;
    const u32 tr_off = (4 * g + q) * (N * 2) + 4 * p * 2;

    const long long wave_tu0 = ((long long)blockIdx.x * 4 + wave) * TPW * a.iters;
    ResFetch<N> nxt;
    if (wave_tu0 < a.n_tu) nxt.issue(a, wave_tu0, lane);
    for (int it = 0; it < a.iters; it++) {
        const long long tu0 = wave_tu0 + (long long)it * TPW;
        if (tu0 >= a.n_tu) break;
        const ResFetch<N> cur = nxt;
        if (it + 1 < a.iters && tu0 + TPW < a.n_tu) nxt.issue(a, tu0 + TPW, lane);
        /* ---- levels (linear 16 B per lane), scaled; transform TUs go to the LDS tile, the others leave ---- */
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const u32 s0 = (64u * c + lane) * CH, tul = s0 / (N * N), pos = s0 % (N * N);
            const u32 inf = cur.inf[c];
            const int qP = inf & 0xff;
            const u32 fl = (inf >> 8) & 0xff, mid = (inf >> 16) & 0xff;
            u32 outd[4];
            if (fl & TU_BYPASS) {
#pragma unroll
                for (int i = 0; i < 4; i++) outd[i] = cur.raw[c][i];
            } else {
                const bool flat = !LISTED || (fl & TU_TSKIP);
                scale_chunk<CH, NARROW>(cur.raw[c], outd, qP, flat, sl + mid * (N * N) + pos, bd_shift, cmin, cmax);
            }
            if (!(fl & (TU_BYPASS | TU_TSKIP))) {
                *(u32x4 *)(tin + s0 * 2) = u32x4{outd[0], outd[1], outd[2], outd[3]};
            } else if (tu0 + tul < a.n_tu) {
                /* hevc.c:4209-4236: r = level, or d << tsShift; optional 180-degree rotation = the chunk reversed at the mirrored position */
                const int ts = (fl & TU_TSKIP) ? 5 + 4 : 0;
                u32 o[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int lo = (int)(short)(outd[i] & 0xffffu) << ts, hi = ((int)outd[i] >> 16) << ts;
                    o[i] = __builtin_amdgcn_perm((u32)hi, (u32)lo, 0x05040100u);
                }
                u32x4 w = {o[0], o[1], o[2], o[3]};
                u32 opos = pos;
                if (fl & TU_ROTATE) {
#pragma unroll
                    for (int i = 0; i < 4; i++) w[i] = __builtin_amdgcn_perm(o[3 - i], o[3 - i], 0x01000302u);
                    opos = N * N - 8 - pos;
                }
                __builtin_nontemporal_store(w, (u32x4 *)(a.res + (tu0 + tul) * (N * N) + opos));
            }
        }
        /* ---- the four TUs, unconditionally (a TU that does not take the transform leaves stale numbers in its
         * part of the output tile, which is then not stored): four independent MFMA chains to interleave.  A TU's
         * flags are wave-uniform here and already in a register: chunk u/2, lanes 32(u&1).. ---- */
        bool xform[TPW];
#pragma unroll
        for (int u = 0; u < TPW; u++) {
            const u32 inf = (u32)__builtin_amdgcn_readlane((int)cur.inf[u >> 1], (u & 1) * 32);
            xform[u] = tu0 + u < a.n_tu && !((inf >> 8) & (TU_BYPASS | TU_TSKIP));
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tin + u * (N * N * 2) + tr_off));
            const u32x2 vv = __builtin_bit_cast(u32x2, v);
            int hi, lo;
            split_bytes(vv[0], vv[1], hi, lo);
            const v4i zero = {};
            v4i eh = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(unsigned long)(u32)hi, mreg, zero, 0, 0, 0);
            v4i el = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(unsigned long)(u32)lo, mreg, c1, 0, 0, 0);
            const u32 g0 = clip_pack<NARROW>(((eh[0] << 8) + el[0]) >> 7, ((eh[1] << 8) + el[1]) >> 7, cmin, cmax);
            const u32 g1 = clip_pack<NARROW>(((eh[2] << 8) + el[2]) >> 7, ((eh[3] << 8) + el[3]) >> 7, cmin, cmax);
            split_bytes(g0, g1, hi, lo);
            eh = __builtin_amdgcn_mfma_i32_16x16x32_i8(mreg, (long)(unsigned long)(u32)hi, zero, 0, 0, 0);
            el = __builtin_amdgcn_mfma_i32_16x16x32_i8(mreg, (long)(unsigned long)(u32)lo, c2, 0, 0, 0);
            /* register i of this lane is r[y = col][x' = 4g + i]: one 8-byte piece of row y */
            u32x2 w;
            w[0] = __builtin_amdgcn_perm((u32)(((eh[1] << 8) + el[1]) >> sh2), (u32)(((eh[0] << 8) + el[0]) >> sh2), 0x05040100u);
            w[1] = __builtin_amdgcn_perm((u32)(((eh[3] << 8) + el[3]) >> sh2), (u32)(((eh[2] << 8) + el[2]) >> sh2), 0x05040100u);
            *(u32x2 *)(tout + (u * N + col) * OST + 4 * g * 2) = w;
        }
        /* ---- out as the same linear copy ---- */
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const u32 s0 = (64u * c + lane) * CH, tul = s0 / (N * N), pos = s0 % (N * N);
            const bool st = tul == 0 ? xform[0] : (tul == 1 ? xform[1] : (tul == 2 ? xform[2] : xform[3]));
            if (st) {
                const char *src = tout + (tul * N + (pos >> 4)) * OST + (pos & 15) * 2;
                const u32x2 w0 = *(const u32x2 *)src, w1 = *(const u32x2 *)(src + 8);
                __builtin_nontemporal_store(u32x4{w0[0], w0[1], w1[0], w1[1]}, (u32x4 *)(a.res + (tu0 + tul) * (N * N) + pos));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * 8x8 TUs on the matrix cores, four to one 16x16 MFMA.  Number the four TUs (p, q), p, q in {0, 1};
 * output row m = (p, x), output column n = (q, y), and the sum runs over 16 slots (q', k):
 *   stage 1  A1[(p,x)][(q',k)] = d_pq'[k][x]            B1[(q',k)][(q,y)] = [q' == q] M8[k][y]
 *   stage 2  A2[(p,x')][(p',x)] = [p' == p] M8[x][x']   B2[(p',x)][(q,y)] = g_p'q[x][y]
 * so C1[(p,x)][(q,y)] = sum_k d_pq[k][x] M8[k][y] and C2[(p,x')][(q,y)] = sum_x M8[x][x'] g_pq[x][y]:
 * all 256 outputs are wanted ones.  C/D keeps rows 4g + i of column n in lane group g, i.e. slot
 * (g >> 1, 4(g & 1) + i): that is the slot order of every operand (bytes 4..7 of each lane, the upper
 * half of the instruction's K = 32, are zero in the data operands), and B1 and A2 are again the same
 * register.  The LDS tile keeps TUs (0, q') and (1, q') side by side as rows of 16 so that one
 * ds_read_b64_tr_b16 per lane fetches the A1 operand.
 * ------------------------------------------------------------------------------------------------ */
struct Mfma8Tab {
    u32 m[64]; /* lane (g, r): byte j = [(r >> 3) == (g >> 1)] M8[4(g & 1) + j][r & 7] */
    int s[8];  /* 128 * sum_k M8[k][i] */
};
constexpr Mfma8Tab make_mfma8_tab()
{
    Mfma8Tab t = {};
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 4; j++) {
            const int g = l >> 4, r = l & 15;
            const int v = (r >> 3) == (g >> 1) ? dct_coef(4 * (4 * (g & 1) + j), r & 7) : 0;
            t.m[l] |= (u32)(uint8_t)(int8_t)v << (8 * j);
        }
    for (int i = 0; i < 8; i++) {
        int sum = 0;
        for (int k = 0; k < 8; k++) sum += dct_coef(4 * k, i);
        t.s[i] = 128 * sum;
    }
    return t;
}
__device__ const Mfma8Tab kTab8 = make_mfma8_tab();

template <bool NARROW, bool LISTED>
__global__ __launch_bounds__(256) void k_hevc_residual8_mfma(HevcResArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t sl[LISTED ? 6 * 8 * 8 : 16];
    if (LISTED) stage_scaling_lists<8>(sl, a.scaling);
    constexpr int N = 8, TPW = 8, OST = 24; /* TUs per batch; byte stride of the output tile's 16-byte rows */
    constexpr int CH = 8;
    __shared__ __attribute__((aligned(16))) char lds_in[4][TPW * N * N * 2];
    __shared__ __attribute__((aligned(16))) char lds_out[4][TPW * N * OST];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *tin = lds_in[wave], *tout = lds_out[wave];
    const u32 g = lane >> 4, r = lane & 15;
    const int range = a.epp ? (a.bitdepth + 6 > 15 ? a.bitdepth + 6 : 15) : 15;
    const int cmin = -(1 << range), cmax = (1 << range) - 1;
    const int bd_shift = a.bitdepth + 3 + 10 - range;
    int sh2 = 20 - a.bitdepth;
    if (a.epp && sh2 < 11) sh2 = 11;
    if (sh2 < 0) sh2 = 0;
    const int rnd2 = sh2 > 0 ? 1 << (sh2 - 1) : 0;
    const long mreg = (long)(unsigned long)kTab8.m[lane];
    v4i c1, c2;
#pragma unroll
    for (int i = 0; i < 4; i++) { c1[i] = kTab8.s[r & 7] + 64; c2[i] = kTab8.s[4 * (g & 1) + i] + rnd2; }
    /* A TU of the batch is t = 4h + 2q + p (h: which MFMA).  Chunk = one row: lane L holds row L & 7 of TU L >> 3,
     * written to [h][q][k][p * 8 ..] */
    const u32 lt = lane >> 3, lk = lane & 7;
    const u32 in_off = ((lt >> 2) * 512) + (((lt >> 1) & 1) * 256) + lk * 32 + (lt & 1) * 16;
    /* transposing read: lane 4qq+pp of 16-lane group g supplies row 4(g & 1) + qq of pair q' = g >> 1, columns 4pp .. */
    const u32 qq = (lane & 15) >> 2, pp = lane & 3;
    const u32 tr_off = (g >> 1) * 256 + (4 * (g & 1) + qq) * 32 + 4 * pp * 2;
    /* result: lane (g, n = (q, y)) holds r[y][x' = 4(g & 1) + i] of TU (p = g >> 1, q): 8 bytes of row y */
    const u32 out_off = ((2 * (r >> 3) + (g >> 1)) * N + (r & 7)) * OST + 4 * (g & 1) * 2; /* + h * 4 * N * OST */
    const u32 lin_out = (lt * N + lk) * OST;

    const long long wave_tu0 = ((long long)blockIdx.x * 4 + wave) * TPW * a.iters;
    ResFetch<N> nxt;
    if (wave_tu0 < a.n_tu) nxt.issue(a, wave_tu0, lane);
    for (int it = 0; it < a.iters; it++) {
        const long long tu0 = wave_tu0 + (long long)it * TPW;
        if (tu0 >= a.n_tu) break;
        const ResFetch<N> cur = nxt;
        if (it + 1 < a.iters && tu0 + TPW < a.n_tu) nxt.issue(a, tu0 + TPW, lane);
        const u32 inf = cur.inf[0];
        const int qP = inf & 0xff;
        const u32 fl = (inf >> 8) & 0xff, mid = (inf >> 16) & 0xff;
        const bool live = tu0 + lt < a.n_tu;
        const bool xf = !(fl & (TU_BYPASS | TU_TSKIP));
        u32 outd[4];
        if (fl & TU_BYPASS) {
#pragma unroll
            for (int i = 0; i < 4; i++) outd[i] = cur.raw[0][i];
        } else {
            const bool flat = !LISTED || (fl & TU_TSKIP);
            scale_chunk<CH, NARROW>(cur.raw[0], outd, qP, flat, sl + mid * (N * N) + lk * 8, bd_shift, cmin, cmax);
        }
        if (xf) {
            *(u32x4 *)(tin + in_off) = u32x4{outd[0], outd[1], outd[2], outd[3]};
        } else if (live) {
            /* hevc.c:4209-4236: r = level, or d << tsShift; optional 180-degree rotation = the row reversed at the mirrored position */
            const int ts = (fl & TU_TSKIP) ? 5 + 3 : 0;
            u32 o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int lo = (int)(short)(outd[i] & 0xffffu) << ts, hi = ((int)outd[i] >> 16) << ts;
                o[i] = __builtin_amdgcn_perm((u32)hi, (u32)lo, 0x05040100u);
            }
            u32x4 w = {o[0], o[1], o[2], o[3]};
            u32 opos = lk * 8;
            if (fl & TU_ROTATE) {
#pragma unroll
                for (int i = 0; i < 4; i++) w[i] = __builtin_amdgcn_perm(o[3 - i], o[3 - i], 0x01000302u);
                opos = N * N - 8 - opos;
            }
            __builtin_nontemporal_store(w, (u32x4 *)(a.res + (tu0 + lt) * (N * N) + opos));
        }
        /* ---- two MFMA groups of four TUs; TUs that do not take the transform leave stale numbers that are not stored ---- */
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tin + h * 512 + tr_off));
            const u32x2 vv = __builtin_bit_cast(u32x2, v);
            int hi, lo;
            split_bytes(vv[0], vv[1], hi, lo);
            const v4i zero = {};
            v4i eh = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(unsigned long)(u32)hi, mreg, zero, 0, 0, 0);
            v4i el = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)(unsigned long)(u32)lo, mreg, c1, 0, 0, 0);
            const u32 g0 = clip_pack<NARROW>(((eh[0] << 8) + el[0]) >> 7, ((eh[1] << 8) + el[1]) >> 7, cmin, cmax);
            const u32 g1 = clip_pack<NARROW>(((eh[2] << 8) + el[2]) >> 7, ((eh[3] << 8) + el[3]) >> 7, cmin, cmax);
            split_bytes(g0, g1, hi, lo);
            eh = __builtin_amdgcn_mfma_i32_16x16x32_i8(mreg, (long)(unsigned long)(u32)hi, zero, 0, 0, 0);
            el = __builtin_amdgcn_mfma_i32_16x16x32_i8(mreg, (long)(unsigned long)(u32)lo, c2, 0, 0, 0);
            u32x2 w;
            w[0] = __builtin_amdgcn_perm((u32)(((eh[1] << 8) + el[1]) >> sh2), (u32)(((eh[0] << 8) + el[0]) >> sh2), 0x05040100u);
            w[1] = __builtin_amdgcn_perm((u32)(((eh[3] << 8) + el[3]) >> sh2), (u32)(((eh[2] << 8) + el[2]) >> sh2), 0x05040100u);
            *(u32x2 *)(tout + h * (4 * N * OST) + out_off) = w;
        }
        if (xf && live) {
            const u32x2 w0 = *(const u32x2 *)(tout + lin_out), w1 = *(const u32x2 *)(tout + lin_out + 8);
            __builtin_nontemporal_store(u32x4{w0[0], w0[1], w1[0], w1[1]}, (u32x4 *)(a.res + (tu0 + lt) * (N * N) + lk * 8));
        }
    }
}

extern "C" int ffhip_hevc_residual_batch(int nTbS, long long n_tu, const int16_t *d_level, const uint8_t *d_tuinfo,
                                         const uint8_t *d_scaling, int bitdepth, int epp, int16_t *d_residual,
                                         void *stream)
{
    if (nTbS != 4 && nTbS != 8 && nTbS != 16 && nTbS != 32) return FFHIP_EINVAL;
    if (n_tu < 0 || bitdepth < 8 || bitdepth > 16) return FFHIP_EINVAL;
    if (n_tu == 0) return FFHIP_OK;
    if (!d_level || !d_tuinfo || !d_residual || ((uintptr_t)d_level & 15) || ((uintptr_t)d_residual & 15) ||
        ((uintptr_t)d_tuinfo & 3) || ((uintptr_t)d_scaling & 3))
        return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    HevcResArgs a = {d_level, d_tuinfo, d_scaling, d_residual, n_tu, bitdepth, epp ? 1 : 0, 1};
    /* the 32x32 TUs take the matrix-core kernel (two TUs per wave); FFHIP_HEVC_RES32=dot keeps them on the butterflies */
    const char *e32 = FFHIP_ENV("FFHIP_HEVC_RES32"), *e16 = FFHIP_ENV("FFHIP_HEVC_RES16"), *e8 = FFHIP_ENV("FFHIP_HEVC_RES8"),
               *e4 = FFHIP_ENV("FFHIP_HEVC_RES4"), *eit = FFHIP_ENV("FFHIP_HEVC_RES_ITERS");
    const bool mfma8 = !(e8 && !strcmp(e8, "dot"));
    const bool mfma16 = !(e16 && !strcmp(e16, "dot"));
    const bool mfma32 = !(e32 && !strcmp(e32, "dot"));
    /* batches of 64 rows per wave: about 2048 samples' worth, fewer when that would leave the chip short of workgroups */
    const int iters_env = eit ? atoi(eit) : 0;
    const bool lane4 = !(e4 && !strcmp(e4, "rows")); /* "rows": the 64-rows-per-wave kernel, for A/B */
    const long long per_batch = nTbS == 32 && mfma32 ? 8 : (nTbS == 4 && lane4 ? 256 : 4LL * (64 / nTbS));
    int iters = iters_env > 0 ? iters_env : (nTbS == 4 ? 8 : (nTbS == 8 ? 4 : 2));
    if (nTbS == 4 && lane4) iters = 1; /* a lane's TUs would be 32 B apart: keep the wave's stream linear */
    while (iters_env <= 0 && iters > 1 && n_tu / (per_batch * iters) < 4096) iters >>= 1;
    a.iters = iters;
    const long long per_wg = per_batch * iters;
    const long long wgs = (n_tu + per_wg - 1) / per_wg;
    if (wgs > 0x7fffffffLL) return FFHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = !epp || bitdepth + 6 <= 15; /* coefficient range is 16 bits: clips are saturating packs */
#define LAUNCH_RES(KERNEL)                                                                      \
    do {                                                                                        \
        if (narrow) hipLaunchKernelGGL((KERNEL<true>), dim3((unsigned)wgs), dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((KERNEL<false>), dim3((unsigned)wgs), dim3(256), 0, st, a);        \
    } while (0)
#define LAUNCH_RESL(KERNEL)                                                                                    \
    do {                                                                                                       \
        if (narrow && d_scaling) hipLaunchKernelGGL((KERNEL<true, true>), dim3((unsigned)wgs), dim3(256), 0, st, a);   \
        else if (narrow) hipLaunchKernelGGL((KERNEL<true, false>), dim3((unsigned)wgs), dim3(256), 0, st, a);          \
        else if (d_scaling) hipLaunchKernelGGL((KERNEL<false, true>), dim3((unsigned)wgs), dim3(256), 0, st, a);       \
        else hipLaunchKernelGGL((KERNEL<false, false>), dim3((unsigned)wgs), dim3(256), 0, st, a);                     \
    } while (0)
#define LAUNCH_RESN(NN)                                                                                    \
    do {                                                                                                   \
        if (narrow) hipLaunchKernelGGL((k_hevc_residual<NN, true>), dim3((unsigned)wgs), dim3(256), 0, st, a);  \
        else hipLaunchKernelGGL((k_hevc_residual<NN, false>), dim3((unsigned)wgs), dim3(256), 0, st, a);        \
    } while (0)
    switch (nTbS) {
    case 4:
        if (lane4) LAUNCH_RES(k_hevc_residual4);
        else LAUNCH_RESN(4);
        break;
    case 8:
        if (mfma8) LAUNCH_RESL(k_hevc_residual8_mfma);
        else LAUNCH_RESN(8);
        break;
    case 16:
        if (mfma16) LAUNCH_RESL(k_hevc_residual16_mfma);
        else LAUNCH_RESN(16);
        break;
    default:
        if (mfma32) LAUNCH_RESL(k_hevc_residual32_mfma);
        else LAUNCH_RESN(32);
        break;
    }
#undef LAUNCH_RES
#undef LAUNCH_RESL
#undef LAUNCH_RESN
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
