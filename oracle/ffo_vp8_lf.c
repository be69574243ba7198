/*
 * oracle/ffo_vp8_lf.c -- CPU restatement of the VP8 in-loop deblocking filter as the
 * reference applies it.  TEST INFRASTRUCTURE ONLY (see oracle/ffo.h).
 *
 * Follows (reference file:line, /root/reference):
 *   DoFilter2/4/6, Hev, NeedsFilter(2)     format/webp.c:1480-1553
 *   Simple{H,V}Filter16(i), FilterLoop24/26 format/webp.c:1555-1626
 *   {H,V}Filter16(i), {H,V}Filter8(i)      format/webp.c:1629-1684
 *   loopfilter (per-MB driver)             format/webp.c:1686-1752
 *   frame loop                             format/webp.c:1856-1866
 * The reference's clip tables (webp.c:200-353) are plain saturations, written as such here.
 * Kept as in the reference: in the NORMAL filter the inner edges are filtered when the
 * macroblock is NOT B_PRED (`if (skip_sub_filter)`, webp.c:1731,1741), in the SIMPLE filter
 * when it IS (`if (!skip_sub_filter)`, webp.c:1710,1720).
 */
#include "ffo.h"

static inline int sclip1(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); } /* VP8ksclip1 */
static inline int sclip2(int v) { return v < -16 ? -16 : (v > 15 ? 15 : v); }     /* VP8ksclip2 */
static inline int clip1(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }       /* VP8kclip1  */
static inline int iabs(int v) { return v < 0 ? -v : v; }

static void filter2(uint8_t *p, int s)
{
    const int p1 = p[-2 * s], p0 = p[-s], q0 = p[0], q1 = p[s];
    const int a = 3 * (q0 - p0) + sclip1(p1 - q1);
    const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3);
    p[-s] = (uint8_t)clip1(p0 + a2);
    p[0] = (uint8_t)clip1(q0 - a1);
}
static void filter4(uint8_t *p, int s)
{
    const int p1 = p[-2 * s], p0 = p[-s], q0 = p[0], q1 = p[s];
    const int a = 3 * (q0 - p0);
    const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3), a3 = (a1 + 1) >> 1;
    p[-2 * s] = (uint8_t)clip1(p1 + a3);
    p[-s] = (uint8_t)clip1(p0 + a2);
    p[0] = (uint8_t)clip1(q0 - a1);
    p[s] = (uint8_t)clip1(q1 - a3);
}
static void filter6(uint8_t *p, int s)
{
    const int p2 = p[-3 * s], p1 = p[-2 * s], p0 = p[-s], q0 = p[0], q1 = p[s], q2 = p[2 * s];
    const int a = sclip1(3 * (q0 - p0) + sclip1(p1 - q1));
    const int a1 = (27 * a + 63) >> 7, a2 = (18 * a + 63) >> 7, a3 = (9 * a + 63) >> 7;
    p[-3 * s] = (uint8_t)clip1(p2 + a3);
    p[-2 * s] = (uint8_t)clip1(p1 + a2);
    p[-s] = (uint8_t)clip1(p0 + a1);
    p[0] = (uint8_t)clip1(q0 - a1);
    p[s] = (uint8_t)clip1(q1 - a2);
    p[2 * s] = (uint8_t)clip1(q2 - a3);
}
static int hev(const uint8_t *p, int s, int t)
{
    return iabs(p[-2 * s] - p[-s]) > t || iabs(p[s] - p[0]) > t;
}
static int needs1(const uint8_t *p, int s, int t) { return 4 * iabs(p[-s] - p[0]) + iabs(p[-2 * s] - p[s]) <= t; }
static int needs2(const uint8_t *p, int s, int t, int it)
{
    if (4 * iabs(p[-s] - p[0]) + iabs(p[-2 * s] - p[s]) > t) return 0;
    return iabs(p[-4 * s] - p[-3 * s]) <= it && iabs(p[-3 * s] - p[-2 * s]) <= it && iabs(p[-2 * s] - p[-s]) <= it &&
           iabs(p[3 * s] - p[2 * s]) <= it && iabs(p[2 * s] - p[s]) <= it && iabs(p[s] - p[0]) <= it;
}

/* one edge of `size` samples: hs = step across the edge, vs = step along it */
static void edge_simple(uint8_t *p, int hs, int vs, int size, int thresh)
{
    for (int i = 0; i < size; i++, p += vs)
        if (needs1(p, hs, 2 * thresh + 1)) filter2(p, hs);
}
static void edge_normal(uint8_t *p, int hs, int vs, int size, int thresh, int ithresh, int hevt, int mb_edge)
{
    for (int i = 0; i < size; i++, p += vs)
        if (needs2(p, hs, 2 * thresh + 1, ithresh)) {
            if (hev(p, hs, hevt)) filter2(p, hs);
            else if (mb_edge) filter6(p, hs);
            else filter4(p, hs);
        }
}

/* webp.c:1686-1752.  filt = {sub_limit, inter_limit, hev_thresh} of the MB's (segment, is_4x4) */
void ffo_vp8_loopfilter_mb(int filter_type, int x, int y, int is_bpred, const uint8_t filt[3], uint8_t *yd, uint8_t *ud,
                           uint8_t *vd, int ys, int uvs)
{
    const int sub = filt[0], inter = filt[1], hevt = filt[2], mb = sub + 4;
    const int skip_sub = !is_bpred;
    if (!sub) return;
    if (filter_type == 1) {
        if (x > 0) edge_simple(yd, 1, ys, 16, mb);
        if (!skip_sub) for (int k = 1; k < 4; k++) edge_simple(yd + 4 * k, 1, ys, 16, sub);
        if (y > 0) edge_simple(yd, ys, 1, 16, mb);
        if (!skip_sub) for (int k = 1; k < 4; k++) edge_simple(yd + 4 * k * ys, ys, 1, 16, sub);
        return;
    }
    if (x > 0) {
        edge_normal(yd, 1, ys, 16, mb, inter, hevt, 1);
        edge_normal(ud, 1, uvs, 8, mb, inter, hevt, 1);
        edge_normal(vd, 1, uvs, 8, mb, inter, hevt, 1);
    }
    if (skip_sub) {
        for (int k = 1; k < 4; k++) edge_normal(yd + 4 * k, 1, ys, 16, sub, inter, hevt, 0);
        edge_normal(ud + 4, 1, uvs, 8, sub, inter, hevt, 0);
        edge_normal(vd + 4, 1, uvs, 8, sub, inter, hevt, 0);
    }
    if (y > 0) {
        edge_normal(yd, ys, 1, 16, mb, inter, hevt, 1);
        edge_normal(ud, uvs, 1, 8, mb, inter, hevt, 1);
        edge_normal(vd, uvs, 1, 8, mb, inter, hevt, 1);
    }
    if (skip_sub) {
        for (int k = 1; k < 4; k++) edge_normal(yd + 4 * k * ys, ys, 1, 16, sub, inter, hevt, 0);
        edge_normal(ud + 4 * uvs, uvs, 1, 8, sub, inter, hevt, 0);
        edge_normal(vd + 4 * uvs, uvs, 1, 8, sub, inter, hevt, 0);
    }
}

/* webp.c:1856-1866.  modes: [n_mb][20] records ([0] intra_y_mode, [18] segment_id);
 * filters: [4 segments][2 (i16, i4x4)][3] = sub_limit, inter_limit, hev_thresh */
void ffo_vp8_loopfilter_frame(int mbcols, int mbrows, int filter_type, const uint8_t *modes, const uint8_t *filters,
                              uint8_t *yp, uint8_t *up, uint8_t *vp)
{
    const int ys = 16 * mbcols, uvs = 8 * mbcols;
    if (filter_type <= 0) return;
    for (int y = 0; y < mbrows; y++)
        for (int x = 0; x < mbcols; x++) {
            const uint8_t *m = modes + 20 * ((long)y * mbcols + x);
            const int bp = m[0] == 4;
            ffo_vp8_loopfilter_mb(filter_type, x, y, bp, filters + ((m[18] & 3) * 2 + bp) * 3, yp + (long)y * 16 * ys + x * 16,
                                  up + (long)y * 8 * uvs + x * 8, vp + (long)y * 8 * uvs + x * 8, ys, uvs);
        }
}
