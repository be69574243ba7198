/*
 * oracle/ffo_hevc_intra.c -- CPU restatement of HEVC intra prediction + reconstruction.
 * TEST INFRASTRUCTURE ONLY (see oracle/ffo.h).
 *
 * Follows (reference file:line, /root/reference):
 *   neighbour gathering            coding/hevc.c:4542-4608  intra_sample_prediction
 *   reference_sample_substitution  coding/hevc.c:4277-4351
 *   filtering_neighbouring_samples coding/hevc.c:4355-4426
 *   hevc_intra_planar / DC / angular  format/predict.c:651-792
 *   rdpcm residual modification    coding/hevc.c:3960-3977
 *   cross-component prediction     coding/hevc.c:3979-3988 (as called at :4750-4756)
 *   construct_pic_pior_to_filtering coding/hevc.c:4252-4274
 *
 * The reference derives neighbour availability from z-scan order, slices and tiles while it
 * parses; here it arrives per TU as bit masks (that is the interface of the batched stage).
 * Unavailable samples start as 0 (the reference's zero-initialised left_default/top_default,
 * hevc.c:4554-4555) and are replaced by the substitution process whenever any is missing.
 */
#include "ffo.h"

#include <string.h>

static int ilog2u(int n) { int l = 0; while (n > 1) { n >>= 1; l++; } return l; }
static int iabs(int a) { return a < 0 ? -a : a; }
static int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* 8.4.4.2.2, hevc.c:4277-4351.  left[0..2n-1], top[-1..2n-1]; un*: 1 = unavailable */
void ffo_hevc_substitute(int16_t *left, int16_t *top, int n, int bitdepth, int n_unavail, const int8_t *unL,
                         const int8_t *unT)
{
    if (n_unavail == 4 * n + 1) {
        for (int i = 0; i < 2 * n; i++) left[i] = (int16_t)(1 << (bitdepth - 1));
        for (int i = -1; i < 2 * n; i++) top[i] = (int16_t)(1 << (bitdepth - 1));
        return;
    }
    if (unL[2 * n - 1]) { /* search upwards along the left column, then along the top row */
        int y;
        for (y = 2 * n - 1; y >= 0; y--)
            if (!unL[y]) { left[2 * n - 1] = left[y]; break; }
        if (y < 0)
            for (int x = -1; x < 2 * n; x++)
                if (!unT[x]) { left[2 * n - 1] = top[x]; break; }
    }
    for (int y = 2 * n - 2; y >= 0; y--)
        if (unL[y]) left[y] = left[y + 1];
    if (unT[-1]) top[-1] = left[0];
    for (int x = 0; x < 2 * n; x++)
        if (unT[x]) top[x] = top[x - 1];
}

/* 8.4.4.2.3, hevc.c:4355-4426 */
void ffo_hevc_filter_neighbours(int16_t *left, int16_t *top, int n, int mode, int cidx, int strong_enabled,
                                int bitdepth_y)
{
    static const int thres[3] = {7, 1, 0};
    if (mode == 1 || n == 4) return;
    int d26 = iabs(mode - 26), d10 = iabs(mode - 10);
    if ((d26 < d10 ? d26 : d10) <= thres[ilog2u(n / 8)]) return;
    int16_t fl[64], ft_[65], *ft = ft_ + 1;
    int bi = strong_enabled && cidx == 0 && n == 32 &&
             iabs(top[-1] + top[2 * n - 1] - 2 * top[n - 1]) < (1 << (bitdepth_y - 5)) &&
             iabs(top[-1] + left[2 * n - 1] - 2 * left[n - 1]) < (1 << (bitdepth_y - 5));
    if (bi) {
        ft[-1] = top[-1];
        for (int i = 0; i < 63; i++) {
            fl[i] = (int16_t)((top[-1] * (63 - i) + (i + 1) * left[63] + 32) >> 6);
            ft[i] = (int16_t)((top[-1] * (63 - i) + (i + 1) * top[63] + 32) >> 6);
        }
        fl[63] = left[63];
        ft[63] = top[63];
    } else {
        ft[-1] = (int16_t)((left[0] + 2 * top[-1] + top[0] + 2) >> 2);
        fl[0] = (int16_t)((left[1] + 2 * left[0] + top[-1] + 2) >> 2);
        for (int y = 1; y < 2 * n - 1; y++) fl[y] = (int16_t)((left[y + 1] + 2 * left[y] + left[y - 1] + 2) >> 2);
        fl[2 * n - 1] = left[2 * n - 1];
        for (int x = 0; x < 2 * n - 1; x++) ft[x] = (int16_t)((top[x - 1] + 2 * top[x] + top[x + 1] + 2) >> 2);
        ft[2 * n - 1] = top[2 * n - 1];
    }
    memcpy(top - 1, ft_, (size_t)(2 * n + 1) * sizeof(int16_t));
    memcpy(left, fl, (size_t)(2 * n) * sizeof(int16_t));
}

/* 8.4.4.2.4-6, predict.c:651-792.  The reference reads the neighbour arrays as uint16_t. */
void ffo_hevc_predict(int mode, const int16_t *left_s, const int16_t *top_s, int n, int cidx, int disable_bf,
                      int dc_filter_disabled, int bitdepth_y, int16_t *pred)
{
    const uint16_t *left = (const uint16_t *)left_s, *top = (const uint16_t *)top_s;
    const int lg = ilog2u(n);
    if (mode == 0) {
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                pred[x + y * n] = (int16_t)(uint16_t)(((n - 1 - x) * left[y] + (x + 1) * top[n] + (n - 1 - y) * top[x] +
                                                       (y + 1) * left[n] + n) >> (lg + 1));
        return;
    }
    if (mode == 1) {
        uint32_t dc = 0;
        for (int i = 0; i < n; i++) dc += left[i] + top[i];
        dc = (dc + (1u << lg)) >> (lg + 1);
        for (int i = 0; i < n * n; i++) pred[i] = (int16_t)(uint16_t)dc;
        if (cidx == 0 && n < 32 && !dc_filter_disabled) {
            pred[0] = (int16_t)(uint16_t)((left[0] + 2 * dc + top[0] + 2) >> 2);
            for (int x = 1; x < n; x++) pred[x] = (int16_t)(uint16_t)((top[x] + 3 * dc + 2) >> 2);
            for (int y = 1; y < n; y++) pred[y * n] = (int16_t)(uint16_t)((left[y] + 3 * dc + 2) >> 2);
        }
        return;
    }
    static const int ang[33] = {32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
    static const int inv[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};
    const int angle = ang[mode - 2];
    int ref_[137], *ref = ref_ + 34;
    const uint16_t *main_ = mode >= 18 ? top : left, *side = mode >= 18 ? left : top;
    /* main[-1] is the corner for both orientations: top[-1] */
    ref[0] = top[-1];
    for (int x = 1; x <= n; x++) ref[x] = main_[x - 1];
    if (angle < 0) {
        if (((n * angle) >> 5) < -1)
            for (int x = -1; x >= ((angle * n) >> 5); x--) {
                int k = (x * inv[mode - 11] + 128) >> 8;
                ref[x] = k == 0 ? top[-1] : side[k - 1];
            }
    } else {
        for (int x = n + 1; x <= 2 * n; x++) ref[x] = main_[x - 1];
    }
    for (int a = 0; a < n; a++) {       /* a: along the prediction direction (row for >= 18, column otherwise) */
        const int idx = ((a + 1) * angle) >> 5, fact = ((a + 1) * angle) & 31;
        for (int b = 0; b < n; b++) {
            int v = fact ? ((32 - fact) * ref[b + idx + 1] + fact * ref[b + idx + 2] + 16) >> 5 : ref[b + idx + 1];
            int x = mode >= 18 ? b : a, y = mode >= 18 ? a : b;
            if (cidx == 0 && n < 32 && !disable_bf) {
                if (mode == 26 && x == 0) v = clip3i(0, (1 << bitdepth_y) - 1, top[0] + ((left[y] - top[-1]) >> 1));
                if (mode == 10 && y == 0) v = clip3i(0, (1 << bitdepth_y) - 1, left[0] + ((top[x] - top[-1]) >> 1));
            }
            pred[x + y * n] = (int16_t)(uint16_t)v;
        }
    }
}

/* 8.6.5, hevc.c:3960-3977.  mdir = predModeIntra / 26.  The horizontal form really is a running
 * sum over the flattened block starting at index n (its x-1 at x = 0 is the previous row's end). */
void ffo_hevc_rdpcm(int mdir, int n, int16_t *r)
{
    if (mdir == 0) {
        for (int i = n; i < n * n; i++) r[i] = (int16_t)(r[i] + r[i - 1]);
    } else {
        for (int y = 1; y < n; y++)
            for (int x = 0; x < n; x++) r[x + n * y] = (int16_t)(r[x + n * y] + r[x + n * (y - 1)]);
    }
}

/* 8.6.6, hevc.c:3979-3988.  Element by element, int arithmetic, int16 store.  The reference's only
 * call site (hevc.c:4753-4755) passes the chroma block for both rY and r; ry is read before r is
 * written, so aliasing is well defined.  Products wrap (oracle flags: -fwrapv). */
void ffo_hevc_cross_component(int res_scale, int n, int bitdepth_y, int bitdepth_c, const int16_t *ry, int16_t *r)
{
    for (int i = 0; i < n * n; i++) {
        const int32_t up = (int32_t)((uint32_t)(int32_t)ry[i] << bitdepth_c) >> bitdepth_y;
        const int32_t prod = (int32_t)((uint32_t)res_scale * (uint32_t)up);
        r[i] = (int16_t)(r[i] + (prod >> 3));
    }
}

/* One TU: gather, substitute, filter, predict, add residual, clip, write (decode_intra_block
 * steps 5-10, hevc.c:4730-4790). */
void ffo_hevc_intra_tu(const ffo_hevc_tu *t, const int16_t *residual, int16_t *plane, int stride, int bitdepth_y,
                       int bitdepth_c)
{
    const int n = 1 << t->log2_size, bd = t->cidx == 0 ? bitdepth_y : bitdepth_c;
    int8_t unL[64] = {0}, unA[65] = {0}, *unT = unA + 1;
    int16_t left[64] = {0}, topa[65] = {0}, *top = topa + 1;
    int un = 0;
    for (int x = -1; x < 2 * n; x++) {
        int ok = x < 0 ? (t->flags & FFO_TU_CORNER) != 0 : (int)((t->avail_top >> x) & 1);
        if (ok) top[x] = plane[(long)(t->y - 1) * stride + t->x + x];
        else { unT[x] = 1; un++; }
    }
    for (int y = 0; y < 2 * n; y++) {
        if ((t->avail_left >> y) & 1) left[y] = plane[(long)(t->y + y) * stride + t->x - 1];
        else { unL[y] = 1; un++; }
    }
    if (un > 0) ffo_hevc_substitute(left, top, n, bd, un, unL, unT);
    if (t->flags & FFO_TU_FILTER)
        ffo_hevc_filter_neighbours(left, top, n, t->pred_mode, t->cidx, (t->flags & FFO_TU_STRONG) != 0, bitdepth_y);
    int16_t pred[32 * 32], res[32 * 32];
    ffo_hevc_predict(t->pred_mode, left, top, n, t->cidx, (t->flags & FFO_TU_NO_BF) != 0,
                     (t->flags & FFO_TU_NO_DC_BF) != 0, bitdepth_y, pred);
    if (t->flags & FFO_TU_RESIDUAL) {
        memcpy(res, residual + t->res_offset, (size_t)n * n * sizeof(int16_t));
        if (t->flags & FFO_TU_RDPCM) ffo_hevc_rdpcm(t->pred_mode / 26, n, res);
        if (t->flags & FFO_TU_CCP) ffo_hevc_cross_component(t->res_scale, n, bitdepth_y, bitdepth_c, res, res);
    } else {
        memset(res, 0, sizeof res);
    }
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            plane[(long)(t->y + y) * stride + t->x + x] = (int16_t)clip3i(0, (1 << bd) - 1, pred[x + y * n] + res[x + y * n]);
}

void ffo_hevc_intra_recon(const ffo_hevc_tu *tus, long n_tus, const int16_t *residual, int16_t *py, int16_t *pu,
                          int16_t *pv, int y_stride, int uv_stride, int bitdepth_y, int bitdepth_c)
{
    for (long i = 0; i < n_tus; i++) {
        const ffo_hevc_tu *t = tus + i;
        ffo_hevc_intra_tu(t, residual, t->cidx == 0 ? py : (t->cidx == 1 ? pu : pv), t->cidx == 0 ? y_stride : uv_stride,
                          bitdepth_y, bitdepth_c);
    }
}
