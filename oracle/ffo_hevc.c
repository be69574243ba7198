/*
 * oracle/ffo_hevc.c -- CPU restatement of the HEVC scaling and inverse
 * transforms.  TEST INFRASTRUCTURE ONLY (see oracle/ffo.h).
 *
 * Follows (reference file:line, /root/reference):
 *   ffo_hevc_idct_4x4_dst  utils/idct.c:9-55       idct_1d_4_16bit, idct_4x4_hevc
 *                          (keeps the `+ (shift-1)` rounding of idct.c:31)
 *   ffo_hevc_scale         coding/hevc.c:3743-3816 scale_transform_coefficients
 *   ffo_hevc_transform     coding/hevc.c:3819-3885 transformation
 *                          coding/hevc.c:3888-3956 transform_scaled_coeffients
 *   ffo_hevc_residual_tu   coding/hevc.c:4209-4236 the bypass / transform-skip / transform
 *                          branches of scale_and_transform
 *
 * Layouts: every block is row-major, d[x + y*nTbS] (x fastest), the layout
 * the reference uses for d[] and r[] (hevc.c:3793,3951).
 */
#include "ffo.h"

static int imax(int a, int b) { return a > b ? a : b; }
static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static int32_t asr32(uint32_t x, int s)
{
    return (x & 0x80000000u) ? (int32_t)~((~x) >> s) : (int32_t)(x >> s);
}
static int ilog2(int n) { int l = 0; while (n > 1) { n >>= 1; l++; } return l; }

/* H.265 8.6.4.2 DST-VII matrix, M[j][i] */
static const int k_dst[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};

/* H.265 8.6.4.2 DCT matrix: M[j][i] = c((2i+1) * j mod 128), c(n) ~ 64*sqrt2*cos(n*pi/64)
 * as tabulated by the standard for n = 0..32, with c(64-n) = -c(n), c(64+n) = -c(n). */
static const int k_cos[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                              61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
static int dct_coef(int j, int i)
{
    if (j == 0) return 64;
    int n = ((2 * i + 1) * j) & 127;
    if (n <= 32) return k_cos[n];
    if (n <= 64) return -k_cos[64 - n];
    if (n <= 96) return -k_cos[n - 64];
    return k_cos[128 - n];
}

static void coeff_range(int bitdepth, int epp, int *cmin, int *cmax)
{
    int e = epp ? imax(15, bitdepth + 6) : 15;
    *cmin = -(1 << e);
    *cmax = (1 << e) - 1;
}

static void dst_1d(const int16_t in[4], int16_t out[4], int cmin, int cmax, int shift)
{
    for (int i = 0; i < 4; i++) {
        uint32_t acc = 0;
        for (int j = 0; j < 4; j++) acc += (uint32_t)(k_dst[j][i] * in[j]);
        /* rounding term is (shift - 1), not 1 << (shift-1): idct.c:31 */
        out[i] = (int16_t)clip3(cmin, cmax, asr32(acc + (uint32_t)(shift - 1), shift));
    }
}

void ffo_hevc_idct_4x4_dst(const int16_t in[16], int16_t out[16], int bitdepth, int epp)
{
    int shift2 = imax(20 - bitdepth, epp ? 11 : 0), cmin, cmax;
    coeff_range(bitdepth, epp, &cmin, &cmax);
    int16_t t[4], e[4][4];
    for (int x = 0; x < 4; x++) { /* columns first */
        for (int y = 0; y < 4; y++) t[y] = in[x + 4 * y];
        dst_1d(t, e[x], cmin, cmax, 7);
    }
    for (int y = 0; y < 4; y++) {
        for (int x = 0; x < 4; x++) t[x] = e[x][y];
        dst_1d(t, out + 4 * y, cmin, cmax, shift2);
    }
}

void ffo_hevc_scale(const int16_t *level, int16_t *d, int nTbS, int qP, int bitdepth, int epp,
                    const uint8_t *scaling_factor)
{
    static const int level_scale[6] = {40, 45, 51, 57, 64, 72};
    int range = epp ? imax(15, bitdepth + 6) : 15, cmin, cmax;
    int bd_shift = bitdepth + ilog2(nTbS) + 10 - range;
    coeff_range(bitdepth, epp, &cmin, &cmax);
    for (int y = 0; y < nTbS; y++)
        for (int x = 0; x < nTbS; x++) {
            uint32_t m = scaling_factor ? scaling_factor[x + y * nTbS] : 16u;
            uint32_t v = (uint32_t)(int32_t)level[x + y * nTbS] * m * (uint32_t)level_scale[qP % 6];
            v <<= (qP / 6);
            v += 1u << (bd_shift - 1);
            d[x + y * nTbS] = (int16_t)clip3(cmin, cmax, asr32(v, bd_shift));
        }
}

static void dct_1d(const int16_t *x, uint32_t *y, int n)
{
    int step = 32 / n; /* matrix row stride, hevc.c:3881 */
    for (int i = 0; i < n; i++) {
        uint32_t acc = 0;
        for (int j = 0; j < n; j++) acc += (uint32_t)(dct_coef(j * step, i) * x[j]);
        y[i] = acc;
    }
}

void ffo_hevc_transform(const int16_t *d, int16_t *r, int nTbS, int trType, int bitdepth, int epp)
{
    if (trType == 1) { /* intra luma 4x4 takes the DST entry point (hevc.c:3907-3921) */
        ffo_hevc_idct_4x4_dst(d, r, bitdepth, epp);
        return;
    }
    int shift2 = imax(20 - bitdepth, epp ? 11 : 0), cmin, cmax;
    coeff_range(bitdepth, epp, &cmin, &cmax);
    int16_t t[32];
    static _Thread_local int16_t g[32][32];
    uint32_t e[32];
    for (int x = 0; x < nTbS; x++) { /* columns, (e + 64) >> 7, clip, int16 */
        for (int y = 0; y < nTbS; y++) t[y] = d[x + y * nTbS];
        dct_1d(t, e, nTbS);
        for (int y = 0; y < nTbS; y++) g[x][y] = (int16_t)clip3(cmin, cmax, asr32(e[y] + 64u, 7));
    }
    for (int y = 0; y < nTbS; y++) { /* rows, rounded shift, no clip, int16 store */
        for (int x = 0; x < nTbS; x++) t[x] = g[x][y];
        dct_1d(t, e, nTbS);
        for (int x = 0; x < nTbS; x++)
            r[x + y * nTbS] = (int16_t)asr32(e[x] + (1u << (shift2 - 1)), shift2);
    }
}

/* One transform unit from levels to residual; flags: 1 = luma intra 4x4 (DST entry),
 * 2 = transform_skip_flag, 4 = cu_transquant_bypass_flag, 8 = rotateCoeffs.
 * scaling_factor: NULL (flat 16) or uint8 [nTbS*nTbS] row-major for this TU's matrix. */
void ffo_hevc_residual_tu(const int16_t *level, int16_t *r, int nTbS, int qP, int flags, int bitdepth, int epp,
                          const uint8_t *scaling_factor)
{
    const int n = nTbS, rot = (flags & 8) != 0;
    if (flags & 4) { /* hevc.c:4209-4222 */
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                r[x + y * n] = rot ? level[(n - x - 1) + (n - y - 1) * n] : level[x + y * n];
        return;
    }
    int16_t d[32 * 32];
    /* scaling lists are not applied to transform-skipped blocks larger than 4x4 (hevc.c:3786-3787) */
    ffo_hevc_scale(level, d, n, qP, bitdepth, epp, ((flags & 2) && n > 4) ? 0 : scaling_factor);
    if (flags & 2) { /* hevc.c:4229-4236 */
        int ts = 5 + ilog2(n);
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                int v = rot ? d[(n - x - 1) + (n - y - 1) * n] : d[x + y * n];
                r[x + y * n] = (int16_t)(uint16_t)((uint32_t)v << ts);
            }
        return;
    }
    ffo_hevc_transform(d, r, n, (flags & 1) && n == 4, bitdepth, epp);
}
