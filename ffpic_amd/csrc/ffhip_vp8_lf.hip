/*
 * ffhip_vp8_lf.hip -- VP8 in-loop deblocking filter for batches of key frames (SURVEY 8f row f3:
 * the step between prediction/reconstruction and colour conversion of the WebP decoder),
 * bit-exact with
 *   DoFilter2/4/6, Hev, NeedsFilter(2)      format/webp.c:1480-1553
 *   Simple*Filter16(i), FilterLoop24/26     format/webp.c:1555-1626
 *   {H,V}Filter16(i), {H,V}Filter8(i)       format/webp.c:1629-1684
 *   loopfilter (per-MB driver, 4 steps)     format/webp.c:1686-1752
 * including the reference's choice of which macroblocks get their inner edges filtered
 * (`if (skip_sub_filter)` in the normal filter, `if (!skip_sub_filter)` in the simple one).
 *
 * Dependency-bound like the predictor: filtering macroblock (x, y) reads and rewrites up to
 * 4 / 3 pixels of its left and top neighbours, which must already be filtered (and (x+1, y-1)
 * must have rewritten the top neighbour's right columns).  Same wavefront levels x + 2y, one
 * launch per level over the batch, one wave per macroblock.  The macroblock and its 4-pixel
 * borders sit in LDS; the vertical-edge phase keeps one pixel row per lane in registers
 * (left edge, then the three inner edges), the horizontal-edge phase one column per lane.
 */
#include "ffhip_internal.h"

#include <vector>

struct Vp8LfArgs {
    const uint8_t *modes;   /* [n_images][n_mb][20]: [0] intra_y_mode, [18] segment_id */
    const uint8_t *filters; /* [4][2][3] sub_limit, inter_limit, hev_thresh             */
    const uint32_t *work;   /* (image, mb) pairs of this level                          */
    uint8_t *y, *u, *v;
    long long plane_y, plane_uv;
    int mbcols, mbrows, count, filter_type;
};

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

#define LS 24 /* luma tile row stride (4 + 16, padded)   */
#define CS 16 /* chroma tile row stride (4 + 8, padded)  */

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_vp8_loopfilter(Vp8LfArgs a)
{
    __shared__ uint8_t tl[4][20 * LS], tc[4][2][12 * CS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + w;
    if (item >= a.count) return;
    const int img = (int)a.work[2 * item], mbi = (int)a.work[2 * item + 1];
    const int n_mb = a.mbcols * a.mbrows, x = mbi % a.mbcols, y = mbi / a.mbcols;
    const uint8_t *m = a.modes + ((long long)img * n_mb + mbi) * 20;
    const bool bpred = m[0] == 4;
    const uint8_t *f = a.filters + ((m[18] & 3) * 2 + (bpred ? 1 : 0)) * 3;
    const int sub = f[0], inter = f[1], hevt = f[2];
    if (!sub) return; /* wave-uniform */
    const int type = a.filter_type;
    /* webp.c:1710-1745: inner edges for B_PRED MBs in the simple filter, for the others in the normal one */
    const bool inner = type == 1 ? bpred : !bpred;
    const int ys = 16 * a.mbcols, us = 8 * a.mbcols;
    uint8_t *Y = a.y + (long long)img * a.plane_y + (long long)y * 16 * ys + x * 16;
    uint8_t *C[2] = {a.u + (long long)img * a.plane_uv + (long long)y * 8 * us + x * 8,
                     a.v + (long long)img * a.plane_uv + (long long)y * 8 * us + x * 8};
    uint8_t *TL = tl[w];
    /* ---- load the macroblock with 4-pixel left/top borders (only what exists) ---- */
    for (int i = lane; i < 20 * 20; i += 64) {
        const int r = i / 20 - 4, c = i % 20 - 4;
        if ((r >= 0 || y > 0) && (c >= 0 || x > 0)) TL[(r + 4) * LS + c + 4] = Y[(long long)r * ys + c];
    }
    if (type != 1)
        for (int i = lane; i < 2 * 144; i += 64) {
            const int pl = i / 144, j = i % 144, r = j / 12 - 4, c = j % 12 - 4;
            if ((r >= 0 || y > 0) && (c >= 0 || x > 0)) tc[w][pl][(r + 4) * CS + c + 4] = C[pl][(long long)r * us + c];
        }
    wave_sync();
    /* ---- vertical edges: one pixel row per lane (lanes 0-15 luma, 16-23 U, 24-31 V) ---- */
    if (lane < 16) {
        int line[20];
        uint8_t *row = TL + (lane + 4) * LS;
#pragma unroll
        for (int k = 0; k < 20; k++) line[k] = row[k];
        filter_line<16>(line, type, x > 0, inner, sub, inter, hevt);
#pragma unroll
        for (int k = 1; k < 19; k++) row[k] = (uint8_t)line[k];
    } else if (lane < 32 && type != 1) {
        int line[12];
        uint8_t *row = tc[w][(lane >> 3) & 1] + ((lane & 7) + 4) * CS;
#pragma unroll
        for (int k = 0; k < 12; k++) line[k] = row[k];
        filter_line<8>(line, type, x > 0, inner, sub, inter, hevt);
#pragma unroll
        for (int k = 1; k < 11; k++) row[k] = (uint8_t)line[k];
    }
    wave_sync();
    /* ---- horizontal edges: one pixel column per lane ---- */
    if (lane < 16) {
        int line[20];
        uint8_t *col = TL + lane + 4;
#pragma unroll
        for (int k = 0; k < 20; k++) line[k] = col[k * LS];
        filter_line<16>(line, type, y > 0, inner, sub, inter, hevt);
#pragma unroll
        for (int k = 1; k < 19; k++) col[k * LS] = (uint8_t)line[k];
    } else if (lane < 32 && type != 1) {
        int line[12];
        uint8_t *col = tc[w][(lane >> 3) & 1] + (lane & 7) + 4;
#pragma unroll
        for (int k = 0; k < 12; k++) line[k] = col[k * CS];
        filter_line<8>(line, type, y > 0, inner, sub, inter, hevt);
#pragma unroll
        for (int k = 1; k < 11; k++) col[k * CS] = (uint8_t)line[k];
    }
    wave_sync();
    /* ---- write back what this macroblock may have changed: its own pixels and the 3 pixels
     * beyond its left / top edge (never the 4x4 corner above-left, which it did not touch) ---- */
    for (int i = lane; i < 19 * 19; i += 64) {
        const int r = i / 19 - 3, c = i % 19 - 3;
        if ((r >= 0 || c >= 0) && (r >= 0 || y > 0) && (c >= 0 || x > 0)) Y[(long long)r * ys + c] = TL[(r + 4) * LS + c + 4];
    }
    if (type != 1)
        for (int i = lane; i < 2 * 121; i += 64) {
            const int pl = i / 121, j = i % 121, r = j / 11 - 3, c = j % 11 - 3;
            if ((r >= 0 || c >= 0) && (r >= 0 || y > 0) && (c >= 0 || x > 0))
                C[pl][(long long)r * us + c] = tc[w][pl][(r + 4) * CS + c + 4];
        }
}

static uint32_t *g_work = nullptr;
static size_t g_work_cap = 0;

extern "C" int ffhip_vp8_loopfilter(int mbcols, int mbrows, int n_images, int filter_type, const uint8_t *d_modes,
                                    const uint8_t *d_filters, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                                    int64_t plane_stride_y, int64_t plane_stride_uv, void *stream)
{
    if (mbcols <= 0 || mbrows <= 0 || n_images < 0 || filter_type < 0 || filter_type > 2) return FFHIP_EINVAL;
    if (n_images == 0 || filter_type == 0) return FFHIP_OK; /* WEBP_FILTER_NONE (webp.c:1852-1856) */
    if (!d_modes || !d_filters || !d_y || !d_u || !d_v) return FFHIP_EINVAL;
    const long long n_mb = (long long)mbcols * mbrows;
    if (n_mb * n_images > 0x3fffffffLL) return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    /* levels x + 2y: the same for every image, no mode-dependent edges here */
    const int n_levels = mbcols + 2 * (mbrows - 1);
    std::vector<std::vector<uint32_t>> lists((size_t)n_levels);
    for (int img = 0; img < n_images; img++)
        for (int y = 0; y < mbrows; y++)
            for (int x = 0; x < mbcols; x++) {
                lists[(size_t)(x + 2 * y)].push_back((uint32_t)img);
                lists[(size_t)(x + 2 * y)].push_back((uint32_t)(y * mbcols + x));
            }
    const size_t total = (size_t)(2 * n_mb * n_images);
    if (total > g_work_cap) {
        if (g_work) (void)hipFree(g_work);
        g_work = nullptr;
        g_work_cap = 0;
        FFHIP_CHECK(hipMalloc((void **)&g_work, total * sizeof(uint32_t)), FFHIP_ENOMEM);
        g_work_cap = total;
    }
    std::vector<uint32_t> flat;
    flat.reserve(total);
    for (auto &l : lists) flat.insert(flat.end(), l.begin(), l.end());
    hipStream_t st = (hipStream_t)stream;
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
    FFHIP_CHECK(hipMemcpy(g_work, flat.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice), FFHIP_EIO);
    Vp8LfArgs a;
    a.modes = d_modes; a.filters = d_filters; a.y = d_y; a.u = d_u; a.v = d_v;
    a.plane_y = plane_stride_y; a.plane_uv = plane_stride_uv;
    a.mbcols = mbcols; a.mbrows = mbrows; a.filter_type = filter_type;
    size_t off = 0;
    for (auto &l : lists) {
        a.work = g_work + off;
        a.count = (int)(l.size() / 2);
        if (a.count) hipLaunchKernelGGL(k_vp8_loopfilter, dim3((unsigned)((a.count + 3) / 4)), dim3(256), 0, st, a);
        off += l.size();
    }
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
