/* ffhip_vp8_filters.h -- the VP8 loop filter's edge filters (format/webp.c:1480-1553) as device functions shared by the row
 * kernels of ffhip_vp8_lf.hip and the fused frame kernel of ffhip_vp8_frame.hip. */
#ifndef FFHIP_VP8_FILTERS_H
#define FFHIP_VP8_FILTERS_H
#include "ffhip_internal.h"

__device__ __forceinline__ int sclip1(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
__device__ __forceinline__ int sclip2(int v) { return v < -16 ? -16 : (v > 15 ? 15 : v); }
__device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

/* p[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 across one edge */
__device__ __forceinline__ void filt2(int *p)
{
    const int a = 3 * (p[4] - p[3]) + sclip1(p[2] - p[5]);
    const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3);
    p[3] = clip255(p[3] + a2);
    p[4] = clip255(p[4] - a1);
}
__device__ __forceinline__ void filt4(int *p)
{
    const int a = 3 * (p[4] - p[3]);
    const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3), a3 = (a1 + 1) >> 1;
    p[2] = clip255(p[2] + a3);
    p[3] = clip255(p[3] + a2);
    p[4] = clip255(p[4] - a1);
    p[5] = clip255(p[5] - a3);
}
__device__ __forceinline__ void filt6(int *p)
{
    const int a = sclip1(3 * (p[4] - p[3]) + sclip1(p[2] - p[5]));
    const int a1 = (27 * a + 63) >> 7, a2 = (18 * a + 63) >> 7, a3 = (9 * a + 63) >> 7;
    p[1] = clip255(p[1] + a3);
    p[2] = clip255(p[2] + a2);
    p[3] = clip255(p[3] + a1);
    p[4] = clip255(p[4] - a1);
    p[5] = clip255(p[5] - a2);
    p[6] = clip255(p[6] - a3);
}
/* one sample position of one edge; s points at p3 of an 8-sample window inside the line */
__device__ __forceinline__ void edge_simple(int *s, int thresh)
{
    if (4 * iabs(s[3] - s[4]) + iabs(s[2] - s[5]) <= 2 * thresh + 1) filt2(s);
}
__device__ __forceinline__ void edge_normal(int *s, int thresh, int ithresh, int hevt, bool mb_edge)
{
    if (4 * iabs(s[3] - s[4]) + iabs(s[2] - s[5]) > 2 * thresh + 1) return;
    if (iabs(s[0] - s[1]) > ithresh || iabs(s[1] - s[2]) > ithresh || iabs(s[2] - s[3]) > ithresh ||
        iabs(s[7] - s[6]) > ithresh || iabs(s[6] - s[5]) > ithresh || iabs(s[5] - s[4]) > ithresh)
        return;
    if (iabs(s[2] - s[3]) > hevt || iabs(s[5] - s[4]) > hevt) filt2(s);
    else if (mb_edge) filt6(s);
    else filt4(s);
}

/* filter all edges that cross one line of N + 4 samples (line[0..3] = the neighbour's last 4) */
template <int N>
__device__ __forceinline__ void filter_line(int *line, int type, bool outer, bool inner, int sub, int inter, int hevt)
{
    const int mb = sub + 4;
    if (outer) {
        if (type == 1) edge_simple(line, mb);
        else edge_normal(line, mb, inter, hevt, true);
    }
    if (inner) {
#pragma unroll
        for (int k = 4; k < N; k += 4) {
            if (type == 1) edge_simple(line + k, sub);
            else edge_normal(line + k, sub, inter, hevt, false);
        }
    }
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* The row kernel filters the 16 luma lines and the 2 x 8 chroma lines of a macroblock with ONE instruction stream: lanes
 * 0-15 hold a luma line (edges at 0, 4, 8, 12), lanes 16-31 a chroma line (edges at 0 and 4; the normal filter only).
 * As two branches of an if / else the chroma lines ran behind the luma lines: half as much again per phase. */
/* The same edge filters without control flow, for the row kernel.  Written as returns and if / else (edge_normal above),
 * every edge became a nest of exec-mask regions around an array of twenty samples, and the compiler merged the arms with
 * register copies: 1 090 of the kernel's 2 080 vector instructions were v_mov.  Here the decision is arithmetic:
 * a filter that must not act gets a = 0, for which every one of its corrections is 0 (sclip2((0 + 4) >> 3) = 0,
 * (27 * 0 + 63) >> 7 = 0, ...), and the two filters an edge chooses between never act together, so their corrections add.
 * filt2 and filt4 are ONE computation: filt4 is filt2 without the p1 - q1 term plus the a3 correction of p1 / q1.
 * Samples are 0..255, so |a - b| is one v_sad_u8. */
__device__ __forceinline__ int absdiff(int a, int b) { return (int)__builtin_amdgcn_sad_u8((unsigned)a, (unsigned)b, 0u); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return __builtin_elementwise_max(__builtin_elementwise_max(a, b), c); }
__device__ __forceinline__ void edge_simple_bf(int *s, int thresh, bool on)
{
    const bool apply = on && 4 * absdiff(s[3], s[4]) + absdiff(s[2], s[5]) <= 2 * thresh + 1;
    const int a = apply ? 3 * (s[4] - s[3]) + sclip1(s[2] - s[5]) : 0;
    const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3);
    s[3] = clip255(s[3] + a2);
    s[4] = clip255(s[4] - a1);
}
template <bool MB_EDGE>
__device__ __forceinline__ void edge_normal_bf(int *s, int thresh, int ithresh, int hevt, bool on)
{
    const int d2 = absdiff(s[2], s[3]), d3 = absdiff(s[5], s[4]);
    const int dmax = __builtin_elementwise_max(max3i(absdiff(s[0], s[1]), absdiff(s[1], s[2]), d2), max3i(absdiff(s[7], s[6]), absdiff(s[6], s[5]), d3));
    const bool apply = on && 4 * absdiff(s[3], s[4]) + absdiff(s[2], s[5]) <= 2 * thresh + 1 && dmax <= ithresh;
    const bool hev = __builtin_elementwise_max(d2, d3) > hevt;
    const int w = sclip1(s[2] - s[5]), base = 3 * (s[4] - s[3]);
    if (MB_EDGE) { /* filt2 where the edge has high variance, filt6 elsewhere */
        const int a = (apply && hev) ? base + w : 0;
        const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3);
        const int b = (apply && !hev) ? sclip1(base + w) : 0;
        const int b1 = (27 * b + 63) >> 7, b2 = (18 * b + 63) >> 7, b3 = (9 * b + 63) >> 7;
        s[1] = clip255(s[1] + b3);
        s[2] = clip255(s[2] + b2);
        s[3] = clip255(s[3] + a2 + b1);
        s[4] = clip255(s[4] - a1 - b1);
        s[5] = clip255(s[5] - b2);
        s[6] = clip255(s[6] - b3);
    } else {       /* filt2 where the edge has high variance, filt4 elsewhere */
        const int a = apply ? base + (hev ? w : 0) : 0;
        const int a1 = sclip2((a + 4) >> 3), a2 = sclip2((a + 3) >> 3), a3 = hev ? 0 : (a1 + 1) >> 1;
        s[2] = clip255(s[2] + a3);
        s[3] = clip255(s[3] + a2);
        s[4] = clip255(s[4] - a1);
        s[5] = clip255(s[5] - a3);
    }
}
/* `outer` (the macroblock has a neighbour on that side) and `lum` (the lane holds a luma line: edges at 8 and 12) are
 * predicates of the edges, not branches around them: a branch around code that rewrites part of a twenty-register array
 * made the compiler keep two copies of the array and move one onto the other where the arms meet (two blocks of sixteen
 * v_mov_b64 per branch).  `inner` stays a branch: it is wave-uniform and skips three edges. */
template <int TYPE>
__device__ __forceinline__ void filter_line_mixed(int *line, bool outer, bool inner, bool lum, int sub, int inter, int hevt)
{
    const int mb = sub + 4;
    if (TYPE == 1) {
        edge_simple_bf(line, mb, outer);
        if (inner) {
            edge_simple_bf(line + 4, sub, true);
            edge_simple_bf(line + 8, sub, lum);
            edge_simple_bf(line + 12, sub, lum);
        }
    } else {
        edge_normal_bf<true>(line, mb, inter, hevt, outer);
        if (inner) {
            edge_normal_bf<false>(line + 4, sub, inter, hevt, true);
            edge_normal_bf<false>(line + 8, sub, inter, hevt, lum);
            edge_normal_bf<false>(line + 12, sub, inter, hevt, lum);
        }
    }
}
/* STRIDE 1: the line is a pixel row (vertical edges); STRIDE LS (== CS): a pixel column (horizontal edges) */
template <int STRIDE, int TYPE>
__device__ __forceinline__ void filter_phase(uint8_t *base, const bool active, const bool lum, bool outer, bool inner, int sub, int inter, int hevt)
{
    if (active) {
        int line[20];
#pragma unroll
        for (int k = 0; k < 20; k++) line[k] = base[k * STRIDE]; /* a chroma lane's k >= 12 reads cells of its own tile array or the bytes behind it: never used */
        filter_line_mixed<TYPE>(line, outer, inner, lum, sub, inter, hevt);
#pragma unroll
        for (int k = 1; k < 11; k++) base[k * STRIDE] = (uint8_t)line[k];
        if (lum) {
#pragma unroll
            for (int k = 11; k < 19; k++) base[k * STRIDE] = (uint8_t)line[k];
        }
    }
}

#endif
