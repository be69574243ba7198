/*
 * oracle/ffo_vp8_pred.c -- CPU restatement of VP8 intra prediction + residual add.
 * TEST INFRASTRUCTURE ONLY (see oracle/ffo.h).
 *
 * Follows (reference file:line, /root/reference):
 *   4x4 luma predictors        format/predict.c:34-249   pred_B_DC .. pred_B_HU
 *   8x8 chroma predictors      format/predict.c:261-306
 *   16x16 luma predictors      format/predict.c:310-355
 *   add_residue_subblock etc.  format/predict.c:378-425
 *   pred_luma                  format/predict.c:426-588  (edge defaults and top-right rules)
 *   pred_chrome                format/predict.c:590-645
 *   frame loop                 format/webp.c:1833-1851   (vp8_decode / vp8_prerdict_mb)
 *
 * Reference behaviours kept on purpose (they differ from the VP8 spec):
 *   - B_PRED top-right of sub-blocks in rows 1-3, column 3 is 127, not the above MB's pixels;
 *   - V_PRED 16x16 copies the memory row above dst (predict.c:338-344), H_PRED 16x16 copies
 *     dst[-1] of every row (predict.c:346-353) -- at the picture's top row / left column these
 *     read bytes outside the macroblock's neighbours.  Here every plane is addressed inside
 *     a caller-provided buffer; bytes before the plane start read as 0.
 */
#include "ffo.h"

#include <string.h>

static inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
#define A3(a, b, c) ((uint8_t)(((uint32_t)(a) + 2u * (uint32_t)(b) + (uint32_t)(c) + 2u) >> 2))
#define A2(a, b) ((uint8_t)(((a) + (b) + 1) >> 1))

/* plane access with "before the plane = 0" (see header) */
static inline int px(const uint8_t *plane, long off) { return off < 0 ? 0 : plane[off]; }

/* 4x4 predictors on an edge array e[0..12] = L K J I X A B C D E F G H
 * (left bottom-to-top, corner, top 8); e[4+1+k] = top[k], e[3-k] = left[k]. */
void ffo_vp8_pred4x4(int mode, const uint8_t e[13], uint8_t out[16])
{
    const uint8_t *top = e + 5, *X = e + 4;
    const uint8_t I = e[3], J = e[2], K = e[1], L = e[0];
#define P(r, c) out[(r) * 4 + (c)]
    switch (mode) {
    case 0: { /* B_DC: predict.c:34-64 */
        int dc = 4;
        for (int i = 0; i < 4; i++) dc += top[i] + e[3 - i];
        memset(out, dc >> 3, 16);
        break;
    }
    case 1: /* B_TM: predict.c:10-18 */
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) P(r, c) = (uint8_t)clamp255(e[3 - r] + top[c] - *X);
        break;
    case 2: /* B_VE */
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) P(r, c) = A3(top[c - 1], top[c], top[c + 1]);
        break;
    case 3: { /* B_HE */
        const uint8_t v[4] = {A3(*X, I, J), A3(I, J, K), A3(J, K, L), A3(K, L, L)};
        for (int r = 0; r < 4; r++) memset(out + 4 * r, v[r], 4);
        break;
    }
    case 4: /* B_RD: along the diagonal d = c - r the value is A3(e[3+d], e[4+d], e[5+d]) */
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) { int d = c - r; P(r, c) = A3(e[3 + d], e[4 + d], e[5 + d]); }
        break;
    case 5: /* B_VR: predict.c:168-195 */
        P(0, 0) = P(2, 1) = A2(*X, top[0]); P(0, 1) = P(2, 2) = A2(top[0], top[1]);
        P(0, 2) = P(2, 3) = A2(top[1], top[2]); P(0, 3) = A2(top[2], top[3]);
        P(1, 0) = P(3, 1) = A3(I, *X, top[0]); P(1, 1) = P(3, 2) = A3(*X, top[0], top[1]);
        P(1, 2) = P(3, 3) = A3(top[0], top[1], top[2]); P(1, 3) = A3(top[1], top[2], top[3]);
        P(3, 0) = A3(I, J, K); P(2, 0) = A3(J, I, *X);
        break;
    case 6: /* B_LD */
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                int s = r + c;
                P(r, c) = s < 6 ? A3(top[s], top[s + 1], top[s + 2]) : A3(top[6], top[7], top[7]);
            }
        break;
    case 7: /* B_VL: predict.c:197-222 */
        P(0, 0) = A2(top[0], top[1]); P(1, 0) = A3(top[0], top[1], top[2]);
        P(2, 0) = P(0, 1) = A2(top[1], top[2]); P(3, 0) = P(1, 1) = A3(top[1], top[2], top[3]);
        P(2, 1) = P(0, 2) = A2(top[2], top[3]); P(3, 1) = P(1, 2) = A3(top[2], top[3], top[4]);
        P(2, 2) = P(0, 3) = A2(top[3], top[4]); P(3, 2) = P(1, 3) = A3(top[3], top[4], top[5]);
        P(2, 3) = A3(top[4], top[5], top[6]); P(3, 3) = A3(top[5], top[6], top[7]);
        break;
    case 8: /* B_HD: predict.c:224-238 */
        P(0, 0) = P(1, 2) = A2(I, *X); P(1, 0) = P(2, 2) = A2(I, J); P(2, 0) = P(3, 2) = A2(J, K); P(3, 0) = A2(K, L);
        P(0, 3) = A3(top[0], top[1], top[2]); P(0, 2) = A3(*X, top[0], top[1]);
        P(0, 1) = P(1, 3) = A3(I, *X, top[0]); P(1, 1) = P(2, 3) = A3(J, I, *X);
        P(2, 1) = P(3, 3) = A3(I, J, K); P(3, 1) = A3(J, K, L);
        break;
    default: /* 9 B_HU: predict.c:240-250 */
        P(0, 0) = A2(I, J); P(0, 1) = A3(I, J, K); P(0, 2) = P(1, 0) = A2(J, K); P(0, 3) = P(1, 1) = A3(J, K, L);
        P(1, 2) = P(2, 0) = A2(K, L); P(1, 3) = P(2, 1) = A3(K, L, L);
        P(2, 2) = P(2, 3) = P(3, 0) = P(3, 1) = P(3, 2) = P(3, 3) = L;
        break;
    }
#undef P
}

/* dst += residual sub-block (16 coefficients, raster), clamped (predict.c:378-389) */
static void add_sub(uint8_t *dst, int stride, const int16_t *c)
{
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++) dst[r * stride + k] = (uint8_t)clamp255(c[4 * r + k] + dst[r * stride + k]);
}

/* DC / TM / VE / HE for an NxN block with edge arrays (top[-1..N-1], left[0..N-1]);
 * the DC rule depends on the MB position (predict.c:261-283, 310-336) */
static void pred_block(int mode, int N, uint8_t *dst, int stride, const uint8_t *top, const uint8_t *left, int x, int y)
{
    int shift = N == 16 ? 4 : 3;
    if (mode == 0) {
        int dc = 0;
        if (x > 0) for (int i = 0; i < N; i++) dc += left[i];
        if (y > 0) for (int i = 0; i < N; i++) dc += top[i];
        if (x == 0 && y == 0) dc = 0x80;
        else if (x == 0 || y == 0) dc = (dc + (1 << (shift - 1))) >> shift;
        else dc = (dc + (1 << shift)) >> (shift + 1);
        for (int r = 0; r < N; r++) memset(dst + r * stride, (uint8_t)dc, N);
    } else if (mode == 1) {
        for (int r = 0; r < N; r++)
            for (int c = 0; c < N; c++) dst[r * stride + c] = (uint8_t)clamp255(left[r] + top[c] - top[-1]);
    } else if (mode == 2) {
        for (int r = 0; r < N; r++) memcpy(dst + r * stride, top, N);
    } else {
        for (int r = 0; r < N; r++) memset(dst + r * stride, left[r], N);
    }
}

/* pred_luma + residual for MB (x, y); plane = whole luma plane, stride = 16*mbcols */
void ffo_vp8_pred_luma(const int16_t *coff, int ymode, const uint8_t imodes[16], uint8_t *plane, int stride, int x,
                       int y)
{
    const long org = (long)y * 16 * stride + x * 16;
    uint8_t *dst = plane + org;
    if (ymode == 4) { /* B_PRED */
        for (int n = 0; n < 16; n++) {
            const int xs = n % 4, ys = n / 4;
            const long so = org + (long)ys * 4 * stride + xs * 4;
            uint8_t e[13]; /* L K J I X A..H */
            for (int m = 0; m < 4; m++) e[3 - m] = (x > 0 || xs > 0) ? plane[so + (long)m * stride - 1] : 129;
            if (y == 0 && ys == 0) {
                memset(e + 4, 127, 9);
            } else if (ys == 0) {
                memcpy(e + 5, plane + so - stride, 8);
                e[4] = (xs > 0 || x > 0) ? plane[so - stride - 1] : 129;
                if (x == stride / 16 - 1 && xs == 3) memset(e + 9, 127, 4);
            } else {
                memcpy(e + 5, plane + so - stride, 4);
                if (xs == 3) memset(e + 9, 127, 4);
                else memcpy(e + 9, plane + so - stride + 4, 4);
                e[4] = (xs == 0 && x == 0) ? 129 : plane[so - stride - 1];
            }
            uint8_t p[16];
            ffo_vp8_pred4x4(imodes[n], e, p);
            for (int r = 0; r < 4; r++) memcpy(plane + so + (long)r * stride, p + 4 * r, 4);
            add_sub(plane + so, stride, coff + 16 * n);
        }
        return;
    }
    uint8_t left[16], topa[17], *top = topa + 1;
    memset(left, 129, 16);
    memset(topa, 127, 17);
    if (x > 0) for (int i = 0; i < 16; i++) left[i] = dst[(long)i * stride - 1];
    if (y > 0) {
        memcpy(top, dst - stride, 16);
        top[-1] = x > 0 ? dst[-stride - 1] : 129;
    }
    if (ymode == 2) { /* V_PRED copies the memory row above, whatever it holds (predict.c:338-344) */
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 16; c++) dst[(long)r * stride + c] = (uint8_t)px(plane, org - stride + c);
    } else if (ymode == 3) { /* H_PRED copies dst[-1] of every row (predict.c:346-353) */
        for (int r = 0; r < 16; r++) memset(dst + (long)r * stride, px(plane, org + (long)r * stride - 1), 16);
    } else {
        pred_block(ymode, 16, dst, stride, top, left, x, y);
    }
    for (int i = 0; i < 4; i++) /* add_luma_block: sub-block (i,j) uses coefficients 16*(4i+j) */
        for (int j = 0; j < 4; j++) add_sub(dst + (long)i * 4 * stride + j * 4, stride, coff + 16 * (4 * i + j));
}

/* pred_chrome + residual for MB (x, y); coff points at the 128 chroma coefficients */
void ffo_vp8_pred_chroma(const int16_t *coff, int mode, uint8_t *uplane, uint8_t *vplane, int stride, int x, int y)
{
    uint8_t *planes[2] = {uplane, vplane};
    for (int pl = 0; pl < 2; pl++) {
        uint8_t *dst = planes[pl] + (long)y * 8 * stride + x * 8;
        uint8_t left[8], topa[9], *top = topa + 1;
        memset(left, 129, 8);
        memset(topa, 127, 9);
        if (x > 0) for (int i = 0; i < 8; i++) left[i] = dst[(long)i * stride - 1];
        if (y > 0) {
            memcpy(top, dst - stride, 8);
            top[-1] = x > 0 ? dst[-stride - 1] : 129;
        }
        pred_block(mode, 8, dst, stride, top, left, x, y);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                add_sub(dst + (long)i * 4 * stride + j * 4, stride, coff + 64 * pl + 16 * (2 * i + j));
    }
}

/* Whole key frame (format/webp.c:1833-1851).  modes: [n_mb][20] = intra_y_mode, intra_uv_mode,
 * imodes[16], 2 pad.  residual: [.][384]; resmap (or NULL): index of the residual row each MB
 * uses -- the reference keeps the previous MB's coefficients for skipped MBs (webp.c:1207-1223). */
void ffo_vp8_recon_frame(int mbcols, int mbrows, const uint8_t *modes, const int16_t *residual, const int32_t *resmap,
                         uint8_t *yp, uint8_t *up, uint8_t *vp)
{
    const int ys = 16 * mbcols, uvs = 8 * mbcols;
    for (int y = 0; y < mbrows; y++)
        for (int x = 0; x < mbcols; x++) {
            const long mb = (long)y * mbcols + x;
            const int16_t *c = residual + 384 * (resmap ? resmap[mb] : mb);
            const uint8_t *m = modes + 20 * mb;
            ffo_vp8_pred_luma(c, m[0], m + 2, yp, ys, x, y);
            ffo_vp8_pred_chroma(c + 256, m[1], up, vp, uvs, x, y);
        }
}
