/*
 * oracle/ffo_vp8.c -- CPU restatement of the VP8 (WebP lossy) residual
 * transforms.  TEST INFRASTRUCTURE ONLY (see oracle/ffo.h).
 *
 * Follows (reference file:line, /root/reference):
 *   ffo_vp8_idct_4x4    utils/idct.c:100-151    idct_4x4_16 (VP8 4x4 IDCT)
 *   ffo_vp8_iwht_long   format/webp.c:1067-1096 IWHT_long
 *   ffo_vp8_iwht_fast   format/webp.c:1098-1106 IWHT_fast
 *   ffo_vp8_residual_mb format/webp.c:1061 (dequant at parse), :1147-1196 (per-MB
 *                       residual assembly in vp8_decode_residual_block)
 */
#include "ffo.h"

/* (x * c) >> 16 with the constants of the VP8 transform; x is an int16 value
 * so the product fits 32 bits. */
#define K1 20091 /* sqrt(2)*cos(pi/8) - 1, Q16; applied as x + (x*K1>>16) */
#define K2 35468 /* sqrt(2)*sin(pi/8), Q16                                  */
static inline int mulk(int x, int k) { return (x * k) >> 16; }

void ffo_vp8_idct_4x4(int16_t blk[16])
{
    int16_t t[16]; /* pass-1 results are stored to int16 (idct.c:124) */
    for (int c = 0; c < 4; c++) { /* vertical pass: column c, rows 0,4,8,12 */
        int x0 = blk[c], x1 = blk[4 + c], x2 = blk[8 + c], x3 = blk[12 + c];
        int s = x0 + x2, d = x0 - x2;
        int lo = mulk(x1, K2) - x3 - mulk(x3, K1);
        int hi = x1 + mulk(x1, K1) + mulk(x3, K2);
        t[c] = (int16_t)(s + hi);
        t[4 + c] = (int16_t)(d + lo);
        t[8 + c] = (int16_t)(d - lo);
        t[12 + c] = (int16_t)(s - hi);
    }
    for (int r = 0; r < 4; r++) { /* horizontal pass, (.. + 4) >> 3 */
        int x0 = t[4 * r], x1 = t[4 * r + 1], x2 = t[4 * r + 2], x3 = t[4 * r + 3];
        int s = x0 + x2, d = x0 - x2;
        int lo = mulk(x1, K2) - x3 - mulk(x3, K1);
        int hi = x1 + mulk(x1, K1) + mulk(x3, K2);
        blk[4 * r + 0] = (int16_t)((s + hi + 4) >> 3);
        blk[4 * r + 1] = (int16_t)((d + lo + 4) >> 3);
        blk[4 * r + 2] = (int16_t)((d - lo + 4) >> 3);
        blk[4 * r + 3] = (int16_t)((s - hi + 4) >> 3);
    }
}

/* Inverse Walsh-Hadamard of the Y2 block; result k = 4*i + j becomes the DC of
 * luma sub-block k, i.e. lands at out[16*k] (webp.c:1091-1094). */
void ffo_vp8_iwht_long(const int16_t in[16], int16_t *out)
{
    int t[16];
    for (int c = 0; c < 4; c++) {
        int a = in[c] + in[12 + c], b = in[4 + c] + in[8 + c];
        int e = in[4 + c] - in[8 + c], f = in[c] - in[12 + c];
        t[c] = a + b;
        t[4 + c] = f + e;
        t[8 + c] = a - b;
        t[12 + c] = f - e;
    }
    for (int r = 0; r < 4; r++) {
        int a = t[4 * r] + t[4 * r + 3], b = t[4 * r + 1] + t[4 * r + 2];
        int e = t[4 * r + 1] - t[4 * r + 2], f = t[4 * r] - t[4 * r + 3];
        out[64 * r + 0] = (int16_t)((a + b + 3) >> 3);
        out[64 * r + 16] = (int16_t)((f + e + 3) >> 3);
        out[64 * r + 32] = (int16_t)((a - b + 3) >> 3);
        out[64 * r + 48] = (int16_t)((f - e + 3) >> 3);
    }
}

void ffo_vp8_iwht_fast(const int16_t in[16], int16_t *out)
{
    int16_t dc = (int16_t)((in[0] + 3) >> 3);
    for (int k = 0; k < 16; k++) out[16 * k] = dc;
}

/* One macroblock from quantised levels to the 384-entry residual the predictor adds.
 *   levels[25][16]  quantised coefficients already at their raster position (the
 *                   zig-zag placement of webp.c:1061 done), blocks 0-15 Y, 16-19 U,
 *                   20-23 V, 24 Y2
 *   nz[25]          the per-block return value of vp8_get_coefficients (tokens read)
 *   has_y2          intra_y_mode != B_PRED
 *   q[6]            y1_dc, y1_ac, y2_dc, y2_ac, uv_dc, uv_ac (struct WEBP_decoder)
 * Dequantised products are stored to int16 (webp.c:1061, `out` is int16_t*); the IDCT
 * of a block runs iff nz > 1 or its DC is non-zero (webp.c:1172,1188) -- so a block
 * whose only token is an AC coefficient keeps that raw value, as in the reference. */
void ffo_vp8_residual_mb(const int16_t *levels, const uint8_t *nz, int has_y2, const uint16_t q[6],
                         int16_t out[384])
{
    for (int i = 0; i < 384; i++) out[i] = 0; /* webp.c:1209 */
    if (has_y2) {
        int16_t dc[16];
        for (int i = 0; i < 16; i++) dc[i] = (int16_t)(levels[24 * 16 + i] * (int)(i ? q[3] : q[2]));
        if (nz[24] > 1) ffo_vp8_iwht_long(dc, out);
        else ffo_vp8_iwht_fast(dc, out);
    }
    for (int b = 0; b < 24; b++) {
        const int dcq = b < 16 ? q[0] : q[4], acq = b < 16 ? q[1] : q[5];
        int16_t *dst = out + 16 * b;
        for (int i = (b < 16 && has_y2) ? 1 : 0; i < 16; i++)
            dst[i] = (int16_t)(levels[b * 16 + i] * (i ? acq : dcq));
        if (nz[b] > 1 || dst[0] != 0) ffo_vp8_idct_4x4(dst);
    }
}
