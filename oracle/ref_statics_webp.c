/*
 * oracle/ref_statics_webp.c -- part of the oracle/_ref build recipe.
 * TEST INFRASTRUCTURE ONLY; compiles only where /root/reference exists.
 *
 * Compiles the reference's own format/webp.c inside this translation unit and
 * exports thin wrappers around its `static` hot-path functions
 * IWHT_long / IWHT_fast (webp.c:1067-1106), its loop filter, and a driver for vp8_decode_residual_block
 * itself (webp.c:1125-1199) over synthetic bool-decoder streams.
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
static void *g_ref_blocks;   /* vp8_decode's per-macroblock header array (webp.c:1824; never freed by the reference) */
static size_t g_ref_blocks_bytes;
static void *ref_malloc_at(size_t n, int line)
{
    if (line == 1824) {
        g_ref_blocks = malloc(n);
        g_ref_blocks_bytes = n;
        return g_ref_blocks;
    }
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
/* segment_id of every macroblock of the last vp8_decode, in raster order, from its own header array */
int ref_webp_segment_ids(uint8_t *out, int max)
{
    const struct macro_block *b = g_ref_blocks;
    const int n = (int)(g_ref_blocks_bytes / sizeof *b);
    for (int i = 0; i < n && i < max; i++) out[i] = b[i].segment_id;
    return n;
}

/* out[0..9] = sharpness_level, segment_feature_mode, lf_update_value[4], loop_filter_adj_enable,
 * mode_ref_lf_delta_update[0], mb_mode_delta_update[0], nbr_partitions: the rest of what
 * calculate_filter_control_parameter reads */
void ref_webp_filter_header(void *wp, int *out)
{
    WEBP *w = wp;
    out[0] = w->k.sharpness_level;
    out[1] = w->k.segmentation.segment_feature_mode;
    for (int s = 0; s < 4; s++) out[2 + s] = w->k.segmentation.lf[s].lf_update_value;
    out[6] = w->k.mb_lf_adjustments.loop_filter_adj_enable;
    out[7] = w->k.mb_lf_adjustments.mode_ref_lf_delta_update[0];
    out[8] = w->k.mb_lf_adjustments.mb_mode_delta_update[0];
    out[9] = w->k.nbr_partitions;
}

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

/* vp8_decode_residual_block ITSELF (webp.c:1125-1199), driven from a synthetic bool-decoder state: the decoder is
 * initialised over caller-provided bytes (any bytes are a valid VP8 token stream) with caller-provided coefficient
 * probabilities, and the reference parses n_mb macroblocks of one row from it: token parse, dequantisation at parse
 * (webp.c:1061), IWHT_long / IWHT_fast by its own "nz > 1" rule, the DC scatter, and idct_4x4 by its own
 * "nz > 1 || dst[0] != 0" rule.  dst_out [n_mb][384] is what it wrote.
 * What the reconstruction stage's caller would hand over for the same macroblocks -- the quantised levels and token
 * counts -- is RECORDED by a second decoder over the same bytes that calls the reference's vp8_get_coefficients in
 * the same order with quantisers 1 (so out[] is the level itself).  The recorder only replays the parse order and
 * contexts; if it replayed them wrongly its levels would not reproduce dst_out through the oracle or the GPU.
 * modes[n_mb]: intra_y_mode per MB (B_PRED = 4 has no Y2 block); seg[n_mb]: segment ids; q: uint16 [4][8] =
 * y1_dc, y1_ac, y2_dc, y2_ac, uv_dc, uv_ac per segment; probs: uint8 [4][8][3][11].
 * levels_out int16 [n_mb][25][16] (block 24 = Y2), nz_out uint8 [n_mb][25].  Returns 0. */
int ref_vp8_residual_blocks_driven(const uint8_t *bytes, int len, int n_mb, const uint8_t *modes, const uint8_t *seg,
                                   const uint16_t *q, const uint8_t *probs, int16_t *levels_out, uint8_t *nz_out,
                                   int16_t *dst_out)
{
    static const int coeff_bands[16] = {0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7};
    WEBP *w = calloc(1, sizeof *w);
    memcpy(w->k.coeff_prob, probs, sizeof w->k.coeff_prob);
    for (int s = 0; s < 4; s++) {
        w->d[s].y1_dc = q[8 * s + 0]; w->d[s].y1_ac = q[8 * s + 1]; w->d[s].y2_dc = q[8 * s + 2];
        w->d[s].y2_ac = q[8 * s + 3]; w->d[s].uv_dc = q[8 * s + 4]; w->d[s].uv_ac = q[8 * s + 5];
    }
    uint8_t *copy_a = malloc((size_t)len + 16), *copy_b = malloc((size_t)len + 16);
    memset(copy_a, 0, (size_t)len + 16);
    memcpy(copy_a, bytes, (size_t)len);
    memcpy(copy_b, copy_a, (size_t)len + 16);
    bool_dec *bt = bool_dec_init(copy_a, len), *rec = bool_dec_init(copy_b, len);
    struct context *top = calloc((size_t)n_mb, sizeof *top), *rtop = calloc((size_t)n_mb, sizeof *rtop);
    struct context left, rleft;
    memset(&left, 0, sizeof left);
    memset(&rleft, 0, sizeof rleft);
    const VP8BandProbas *bands[NUM_TYPES][16];
    for (int t = 0; t < NUM_TYPES; ++t)
        for (int b = 0; b < 16; ++b) bands[t][b] = &w->k.coeff_prob[t][coeff_bands[b]];
    memset(levels_out, 0, (size_t)n_mb * 25 * 16 * sizeof(int16_t));
    memset(nz_out, 0, (size_t)n_mb * 25);
    for (int x = 0; x < n_mb; x++) {
        struct macro_block b;
        memset(&b, 0, sizeof b);
        b.intra_y_mode = modes[x];
        b.segment_id = seg[x] & 3;
        b.x = x;
        int16_t *dst = dst_out + 384 * x;
        memset(dst, 0, 384 * sizeof(int16_t)); /* vp8_decode zeroes the MB's coefficients before the call (webp.c:1844 area) */
        vp8_decode_residual_block(w, &b, dst, &left, top, bt);
        /* ---- the recorder: same parse, quantisers 1 ---- */
        int16_t *lv = levels_out + (size_t)x * 25 * 16;
        uint8_t *nz = nz_out + (size_t)x * 25;
        int first = 0;
        const VP8BandProbas *const *acp = bands[3];
        if (b.intra_y_mode != B_PRED) {
            const int ctx = rtop[x].ctx[0] + rleft.ctx[0];
            const int n = vp8_get_coefficients(rec, lv + 24 * 16, bands[1], 0, ctx, 1, 1);
            rtop[x].ctx[0] = rleft.ctx[0] = n > 0;
            nz[24] = (uint8_t)n;
            first = 1;
            acp = bands[0];
        }
        for (int y = 0; y < 4; y++) {
            uint8_t l = rleft.ctx[y + 1];
            for (int xx = 0; xx < 4; xx++) {
                const int n = vp8_get_coefficients(rec, lv + (y * 4 + xx) * 16, acp, first, rtop[x].ctx[xx + 1] + l, 1, 1);
                nz[y * 4 + xx] = (uint8_t)n;
                l = rtop[x].ctx[xx + 1] = n > 0;
            }
            rleft.ctx[y + 1] = l;
        }
        int blk = 16;
        for (int ch = 5; ch <= 7; ch += 2)
            for (int y = 0; y < 2; y++) {
                uint8_t l = rleft.ctx[y + ch];
                for (int xx = 0; xx < 2; xx++, blk++) {
                    const int n = vp8_get_coefficients(rec, lv + blk * 16, bands[2], 0, l + rtop[x].ctx[xx + ch], 1, 1);
                    nz[blk] = (uint8_t)n;
                    l = rtop[x].ctx[xx + ch] = n > 0;
                }
                rleft.ctx[y + ch] = l;
            }
    }
    /* both decoders must have consumed the same bits: the recorder replayed the reference's parse */
    const int same = bt->value == rec->value && bt->range == rec->range && bt->count == rec->count;
    const long used = (long)(bt->bits->ptr - bt->bits->start);
    bool_dec_free(bt); /* frees the byte buffers too (utils/bitstream.c:40-45) */
    bool_dec_free(rec);
    free(top); free(rtop); free(w);
    if (used + 8 >= len) return -2; /* the caller's bytes ran out: everything behind that point would be padding */
    return same ? 0 : -1;
}

/* calculate_filter_control_parameter (webp.c:1756-1803) on a zero-initialised decoder with the given frame-header
 * fields, looped as WEBP_read_frame does (webp.c:1905-1915: over the DCT partition index, hdr[12], capped at 4 here).
 * hdr: filter_type, loop_filter_level, sharpness_level, segmentation_enabled, segment_feature_mode,
 * lf_update_value[4], loop_filter_adj_enable, mode_ref_lf_delta_update[0], mb_mode_delta_update[0], nbr_partitions (13 ints).
 * out[24] = filters[4][2]{sub_limit, inter_limit, hev_thresh}. */
void ref_webp_filter_params(const int *hdr, int *out)
{
    WEBP *w = calloc(1, sizeof *w);
    w->k.filter_type = hdr[0];
    w->k.loop_filter_level = hdr[1];
    w->k.sharpness_level = hdr[2];
    w->k.segmentation.segmentation_enabled = hdr[3];
    w->k.segmentation.segment_feature_mode = hdr[4];
    for (int s = 0; s < 4; s++) w->k.segmentation.lf[s].lf_update_value = hdr[5 + s];
    w->k.mb_lf_adjustments.loop_filter_adj_enable = hdr[9];
    w->k.mb_lf_adjustments.mode_ref_lf_delta_update[0] = hdr[10];
    w->k.mb_lf_adjustments.mb_mode_delta_update[0] = hdr[11];
    for (int i = 0; i < hdr[12] && i < 4; i++) {
        calculate_filter_control_parameter(w, i, 0);
        calculate_filter_control_parameter(w, i, 1);
    }
    for (int s = 0; s < 4; s++)
        for (int k = 0; k < 2; k++) {
            out[(s * 2 + k) * 3] = w->filters[s][k].sub_limit;
            out[(s * 2 + k) * 3 + 1] = w->filters[s][k].inter_limit;
            out[(s * 2 + k) * 3 + 2] = w->filters[s][k].hev_thresh;
        }
    free(w);
}

/* ---- config 4's CPU baseline (bench.py `extra.c4.cpu_baseline`): ONE key frame through the four post-entropy stages in C,
 * one call per frame, every arithmetic step the reference's own code -- per macroblock the dequantising stores, IWHT_long /
 * IWHT_fast and get_dct_ops(16)->idct_4x4 under the control flow of vp8_decode_residual_block (ref_vp8_residual_mb above),
 * pred_luma / pred_chrome, then the frame's loopfilter() pass and YUV420_to_BGRA32, looped like vp8_decode
 * (webp.c:1833-1868).  levels [n_mb][25][16], info [n_mb][32] (token counts 0..24, [25] has_y2, [26] segment), quant [4][8];
 * residual [n_mb][384] is scratch that also comes back; the planes are preceded by one readable row (predict.c:338-353). */
void ref_vp8_chain_frame(int mbcols, int mbrows, const int16_t *levels, const uint8_t *info, const uint16_t *quant,
                         const uint8_t *modes, int filter_type, const uint8_t *filters, int16_t *residual, uint8_t *yp,
                         uint8_t *up, uint8_t *vp, uint8_t *bgra, int pitch)
{
    const long n_mb = (long)mbcols * mbrows;
    for (long i = 0; i < n_mb; i++)
        ref_vp8_residual_mb(levels + i * 400, info + i * 32, info[i * 32 + 25], quant + 8 * (info[i * 32 + 26] & 3), residual + i * 384);
    ref_vp8_recon_frame(mbcols, mbrows, modes, residual, NULL, yp, up, vp);
    if (filter_type) ref_vp8_loopfilter_frame(mbcols, mbrows, filter_type, modes, filters, yp, up, vp);
    YUV420_to_BGRA32(bgra, pitch, yp, up, vp, 16 * mbcols, 8 * mbcols, mbrows, mbcols);
}
