/*
 * oracle/ref_statics_webp.c -- part of the oracle/_ref build recipe.
 * TEST INFRASTRUCTURE ONLY; compiles only where /root/reference exists.
 *
 * Compiles the reference's own format/webp.c inside this translation unit and
 * exports thin wrappers around its `static` hot-path functions
 * IWHT_long / IWHT_fast (webp.c:1067-1106).
 */
#include <stdlib.h>
#include <string.h>

/* The reference's 16x16 V_PRED / H_PRED read raw memory above / left of the luma plane
 * (format/predict.c:338-353) and its planes come from plain malloc (webp.c:1818-1821), so what a
 * whole-file decode shows at the top row / left column depends on heap contents.  To make the
 * reference deterministic for fixture generation, the picture planes allocated by vp8_decode
 * are zero-filled and preceded by 8 KiB of zeros -- the same convention the batched API documents
 * ("bytes before a plane read as 0").  Pure build-recipe plumbing: no reference source is changed. */
#define REF_GUARD 8192
/* only the picture planes of vp8_decode (webp.c:1819-1822: w->data, Y, U, V) are guarded; every
 * other allocation of the file keeps the real malloc/free because other translation units free
 * them (the bool decoder's buffers, webp.c:1890,1906 -> coding/booldec.c) */
static void *ref_malloc_at(size_t n, int line)
{
    if (line >= 1819 && line <= 1822) {
        char *p = calloc(1, n + REF_GUARD);
        return p ? p + REF_GUARD : NULL;
    }
    return malloc(n);
}
static void ref_free_at(void *p, int line)
{
    if (line == 2088) { /* WEBP_free: free(w->data) */
        if (p) free((char *)p - REF_GUARD);
        return;
    }
    free(p);
}
#define malloc(n) ref_malloc_at(n, __LINE__)
#define free(p) ref_free_at(p, __LINE__)

#include "webp.c" /* the reference's format/webp.c */

void ref_vp8_iwht_long(const int16_t *in, int16_t *out) { IWHT_long(in, out); }
void ref_vp8_iwht_fast(int16_t *in, int16_t *out) { IWHT_fast(in, out); }
void ref_vp8_idct_4x4(int16_t *blk) { get_dct_ops(16)->idct_4x4(blk, 8); }

/* Per-MB residual assembly with the control flow of vp8_decode_residual_block
 * (webp.c:1147-1196) but fed from already-parsed levels instead of the bool decoder:
 * every arithmetic step is the reference's own code (the int16 store of the dequantised
 * product at webp.c:1061, IWHT_long / IWHT_fast, get_dct_ops(16)->idct_4x4). */
void ref_vp8_residual_mb(const int16_t *levels, const uint8_t *nz, int has_y2, const uint16_t *q, int16_t *out)
{
    const struct dct_ops *dct = get_dct_ops(16);
    memset(out, 0, 384 * sizeof(int16_t));
    int16_t *dst = out;
    if (has_y2) {
        int16_t dc[16] = {0};
        for (int n = 0; n < 16; n++) dc[n] = levels[24 * 16 + n] * (n > 0 ? q[3] : q[2]);
        if (nz[24] > 1) IWHT_long(dc, dst);
        else IWHT_fast(dc, dst);
    }
    for (int b = 0; b < 24; b++) {
        const int quant_dc = b < 16 ? q[0] : q[4], quant_ac = b < 16 ? q[1] : q[5];
        for (int n = (b < 16 && has_y2) ? 1 : 0; n < 16; n++) dst[n] = levels[b * 16 + n] * (n > 0 ? quant_ac : quant_dc);
        if (nz[b] > 1 || dst[0] != 0) dct->idct_4x4(dst, 8);
        dst += 16;
    }
}

/* Whole key-frame prediction + reconstruction through the reference's exported pred_luma /
 * pred_chrome (format/predict.c:426-645), looped like vp8_decode (webp.c:1833-1851).
 * The planes must be preceded by at least one readable row (the reference's 16x16 V_PRED /
 * H_PRED read dst - stride and dst[-1], predict.c:338-353). */
void ref_vp8_recon_frame(int mbcols, int mbrows, const uint8_t *modes, int16_t *residual, const int32_t *resmap,
                         uint8_t *yp, uint8_t *up, uint8_t *vp)
{
    const int y_stride = 16 * mbcols, uv_stride = 8 * mbcols;
    for (int y = 0; y < mbrows; y++)
        for (int x = 0; x < mbcols; x++) {
            const long mb = (long)y * mbcols + x;
            int16_t *coeffs = residual + 384 * (resmap ? resmap[mb] : mb);
            uint8_t imodes[16];
            memcpy(imodes, modes + 20 * mb + 2, 16);
            pred_luma(coeffs, modes[20 * mb], imodes, yp + y_stride * y * 16 + x * 16, y_stride, x, y);
            pred_chrome(coeffs + 256, modes[20 * mb + 1], up + 8 * uv_stride * y + x * 8,
                        vp + 8 * uv_stride * y + x * 8, uv_stride, x, y);
        }
}

/* Whole-frame in-loop filter through the reference's static loopfilter() (webp.c:1686-1752),
 * looped like vp8_decode (webp.c:1856-1866).  filters: [4][2][3] = sub_limit, inter_limit,
 * hev_thresh as calculate_filter_control_parameter (webp.c:1756-1803) would have left them. */
void ref_vp8_loopfilter_frame(int mbcols, int mbrows, int filter_type, const uint8_t *modes, const uint8_t *filters,
                              uint8_t *yp, uint8_t *up, uint8_t *vp)
{
    WEBP *w = calloc(1, sizeof *w);
    const int y_stride = 16 * mbcols, uv_stride = 8 * mbcols;
    for (int s = 0; s < 4; s++)
        for (int k = 0; k < 2; k++) {
            w->filters[s][k].sub_limit = filters[(s * 2 + k) * 3];
            w->filters[s][k].inter_limit = filters[(s * 2 + k) * 3 + 1];
            w->filters[s][k].hev_thresh = filters[(s * 2 + k) * 3 + 2];
        }
    for (int y = 0; y < mbrows; y++)
        for (int x = 0; x < mbcols; x++) {
            struct macro_block b;
            memset(&b, 0, sizeof b);
            b.intra_y_mode = modes[20 * (y * mbcols + x)];
            b.segment_id = modes[20 * (y * mbcols + x) + 18] & 3;
            b.x = x;
            loopfilter(w, &b, filter_type, y, yp + y_stride * y * 16 + x * 16, up + 8 * uv_stride * y + x * 8,
                       vp + 8 * uv_stride * y + x * 8, y_stride, uv_stride);
        }
    free(w);
}

/* Loop-filter state of a decoded WEBP (struct pic.pic of WEBP_load, webp.c:2002-2005): out[0] =
 * loop_filter_level, out[1] = filter_type bit, out[2] = segmentation_enabled, out[3..26] =
 * filters[4][2] {sub_limit, inter_limit, hev_thresh} (webp.c:1756-1803). */
void ref_webp_filter_info(void *wp, int *out)
{
    WEBP *w = wp;
    out[0] = w->k.loop_filter_level;
    out[1] = w->k.filter_type;
    out[2] = w->k.segmentation.segmentation_enabled;
    for (int s = 0; s < 4; s++)
        for (int k = 0; k < 2; k++) {
            out[3 + (s * 2 + k) * 3] = w->filters[s][k].sub_limit;
            out[4 + (s * 2 + k) * 3] = w->filters[s][k].inter_limit;
            out[5 + (s * 2 + k) * 3] = w->filters[s][k].hev_thresh;
        }
}
