/*
 * ffhip_hevc.hip -- HEVC residual stage for batches of transform units of one size:
 * scaling (dequantisation) and the 2-D inverse transform, bit-exact with
 *   scale_transform_coefficients   coding/hevc.c:3743-3816
 *   transformation                 coding/hevc.c:3819-3885 (matrix :3826-3859)
 *   transform_scaled_coeffients    coding/hevc.c:3888-3956
 *   idct_4x4_hevc (luma intra 4x4) utils/idct.c:9-55, with its `+ (shift-1)` rounding
 *   transform-skip / bypass glue   coding/hevc.c:4209-4236
 *
 * HBM-bound: 2 B of levels in + 2 B of residual out per sample.  A wave holds 64/N TUs = 64 rows.
 * Levels come in and residuals go out as linear 16-byte-per-lane copies through the wave's LDS tile
 * (a row per lane would make every load a 64-lane gather with a 2N-byte stride: 1.5 TB/s at N = 32);
 * in between one lane owns one column (then one row) for the two 1-D passes.  The transposes go through LDS with ds_read_b64_tr_b16, whose
 * row order is the bit-reversal-like order of the partial butterflies so each lane
 * receives ready-made (x_a, x_b) int16 pairs for v_dot2_i32_i16; the N-point transform is
 * the recursive even/odd decomposition of the H.265 matrix (N/2-point on the even inputs
 * plus an N/2 x N/2 odd part), all accumulated mod 2^32 like the reference's int.
 * No MFMA: fixed small integer transforms.
 */
#include "ffhip_internal.h"

#define PK16(lo, hi) ((u32)(uint16_t)(int16_t)(lo) | ((u32)(uint16_t)(int16_t)(hi) << 16))

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int dot2(u32 a, u32 b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
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
            int o = 0;
#pragma unroll
            for (int k = 0; k < N / 4; k++) /* odd inputs x[4k+1], x[4k+3] */
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
        const int e0 = dot2(in[0], PK16(64, 64), rnd), e1 = dot2(in[0], PK16(64, -64), rnd);
        const int o0 = dot2(in[1], PK16(83, 36), 0), o1 = dot2(in[1], PK16(36, -83), 0);
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
};

#define TU_DST 1u    /* trType 1: luma intra 4x4 -> idct_4x4_hevc                     */
#define TU_TSKIP 2u  /* transform_skip_flag: r = d << (5 + log2 N)                    */
#define TU_BYPASS 4u /* cu_transquant_bypass_flag: r = level                          */
#define TU_ROTATE 8u /* rotateCoeffs (4x4 intra with transform_skip_rotation_enabled) */

__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

template <int N>
__global__ __launch_bounds__(256) void k_hevc_residual(HevcResArgs a)
{
    constexpr int W = N / 2;          /* dwords per row                       */
    constexpr int TPW = 64 / N;       /* TUs per wave                         */
    constexpr int LOG2N = N == 4 ? 2 : (N == 8 ? 3 : (N == 16 ? 4 : 5));
    __shared__ __attribute__((aligned(16))) char lds_all[4 * 64 * N * 2];
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *tile = lds_all + wave * (64 * N * 2);
    const u32 tu_l = lane / N, idx = lane % N; /* TU within the wave, row (load/store) or column/row (passes) */
    long long tu = ((long long)blockIdx.x * 4 + wave) * TPW + tu_l;
    const bool live = tu < a.n_tu;
    if (!live) tu = a.n_tu - 1; /* keep EXEC full for the transposing reads; stores are masked */

    const u32 info = *(const u32 *)(a.tuinfo + tu * 4);
    const u32 flags = (info >> 8) & 0xff;
    const int range = a.epp ? (a.bitdepth + 6 > 15 ? a.bitdepth + 6 : 15) : 15;
    const int cmin = -(1 << range), cmax = (1 << range) - 1;

    /* ---- levels -> scaled coefficients d (hevc.c:3786-3805) -> LDS, as a LINEAR copy: lane L of pass c
     * takes samples (64c + L)*CH .. +CH of the wave's 64 rows, so every load is 16 (8) contiguous bytes
     * per lane and the tile in LDS is simply row-major [tu][row][col] ---- */
    constexpr int CH = N >= 8 ? 8 : 4;  /* samples per lane per pass */
    constexpr int NCH = N / CH;         /* passes                    */
    const long long tu0 = ((long long)blockIdx.x * 4 + wave) * TPW;
    bool chunk_store[NCH];              /* does this chunk's TU take the transform path and exist? */
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u32 s0 = (64u * c + lane) * CH, tul = s0 / (N * N), pos = s0 % (N * N);
        long long tuc = tu0 + tul;
        const bool livec = tuc < a.n_tu;
        if (!livec) tuc = a.n_tu - 1;
        const u32 inf = *(const u32 *)(a.tuinfo + tuc * 4);
        const int qP = inf & 0xff;
        const u32 fl = (inf >> 8) & 0xff, mid = (inf >> 16) & 0xff;
        chunk_store[c] = livec && !(fl & (TU_BYPASS | TU_TSKIP));
        u32 raw[CH / 2];
        if (CH == 8) {
            const u32x4 v = __builtin_nontemporal_load((const u32x4 *)(a.level + tuc * (N * N) + pos));
            raw[0] = v[0]; raw[1] = v[1]; raw[CH / 2 - 2] = v[2]; raw[CH / 2 - 1] = v[3];
        } else {
            const u32x2 v = __builtin_nontemporal_load((const u32x2 *)(a.level + tuc * (N * N) + pos));
            raw[0] = v[0]; raw[1] = v[1];
        }
        u32 outd[CH / 2];
        if (fl & TU_BYPASS) {
#pragma unroll
            for (int i = 0; i < CH / 2; i++) outd[i] = raw[i];
        } else {
            const int ls = qP % 6 == 0 ? 40 : (qP % 6 == 1 ? 45 : (qP % 6 == 2 ? 51 : (qP % 6 == 3 ? 57 : (qP % 6 == 4 ? 64 : 72))));
            const int sh = qP / 6;
            const int bd_shift = a.bitdepth + LOG2N + 10 - range;
            const u32 rnd = 1u << (bd_shift - 1);
            const bool flat = a.scaling == nullptr || ((fl & TU_TSKIP) && N > 4);
            u32 mraw[CH / 4] = {};
            if (!flat) {
#pragma unroll
                for (int i = 0; i < CH / 4; i++) mraw[i] = *(const u32 *)(a.scaling + mid * (N * N) + pos + 4 * i);
            }
            int dv[CH];
#pragma unroll
            for (int i = 0; i < CH; i++) {
                const int lv = (i & 1) ? (int)raw[i >> 1] >> 16 : (int)(short)(raw[i >> 1] & 0xffffu);
                const u32 m = flat ? 16u : (mraw[i >> 2] >> (8 * (i & 3))) & 0xffu;
                u32 v = (u32)lv * m * (u32)ls;
                v <<= sh;
                v += rnd;
                dv[i] = (int)(short)clip3i(cmin, cmax, (int)v >> bd_shift);
            }
#pragma unroll
            for (int i = 0; i < CH / 2; i++) outd[i] = ((u32)dv[2 * i] & 0xffffu) | ((u32)dv[2 * i + 1] << 16);
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
    u32 *trow = (u32 *)(tile + (tu_l * N + idx) * (N * 2)); /* this lane's row of its TU */
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

    /* transposing reads: 16-lane group gq, lane 4q+p supplies row order[4k+q], 4 columns */
    const u32 t16 = lane & 15, q = t16 >> 2, p = t16 & 3, g16 = lane >> 4;
    u32 col_tu, col_x0;
    if (N >= 16) { col_tu = g16 / (N / 16 > 0 ? N / 16 : 1); col_x0 = 16 * (g16 % (N / 16 > 0 ? N / 16 : 1)) + 4 * p; }
    else if (N == 8) { col_tu = 2 * g16 + (p >> 1); col_x0 = 4 * (p & 1); }
    else { col_tu = 4 * g16 + p; col_x0 = 0; }
    const char *tr_base = tile + col_tu * (N * N * 2) + col_x0 * 2;

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
        const int g0 = clip3i(cmin, cmax, e[2 * i] >> 7), g1 = clip3i(cmin, cmax, e[2 * i + 1] >> 7);
        trow[i] = ((u32)g0 & 0xffffu) | ((u32)g1 << 16);
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
    for (int i = 0; i < W; i++) trow[i] = ((u32)r[2 * i] & 0xffffu) | ((u32)r[2 * i + 1] << 16);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u32 s0 = (64u * c + lane) * CH;
        if (chunk_store[c]) {
            int16_t *dstp = a.res + tu0 * (N * N) + s0;
            if (CH == 8) __builtin_nontemporal_store(*(const u32x4 *)(tile + s0 * 2), (u32x4 *)dstp);
            else __builtin_nontemporal_store(*(const u32x2 *)(tile + s0 * 2), (u32x2 *)dstp);
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
    HevcResArgs a = {d_level, d_tuinfo, d_scaling, d_residual, n_tu, bitdepth, epp ? 1 : 0};
    const long long per_wg = 4LL * (64 / nTbS);
    const long long wgs = (n_tu + per_wg - 1) / per_wg;
    if (wgs > 0x7fffffffLL) return FFHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (nTbS) {
    case 4: hipLaunchKernelGGL(k_hevc_residual<4>, dim3((unsigned)wgs), dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL(k_hevc_residual<8>, dim3((unsigned)wgs), dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL(k_hevc_residual<16>, dim3((unsigned)wgs), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(k_hevc_residual<32>, dim3((unsigned)wgs), dim3(256), 0, st, a); break;
    }
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
