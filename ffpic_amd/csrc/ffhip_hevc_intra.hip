/*
 * ffhip_hevc_intra.hip -- HEVC intra prediction + reconstruction for lists of transform
 * units, bit-exact with
 *   intra_sample_prediction        coding/hevc.c:4542-4662 (neighbour gathering :4570-4608)
 *   reference_sample_substitution  coding/hevc.c:4277-4351
 *   filtering_neighbouring_samples coding/hevc.c:4355-4426
 *   hevc_intra_planar / DC / angular  format/predict.c:651-792
 *   residual_modification_transform_bypass (rdpcm)  coding/hevc.c:3960-3977
 *   residual_modification_transform_cross_prediction coding/hevc.c:3979-3988 (as called at :4750-4756)
 *   construct_pic_pior_to_filtering coding/hevc.c:4252-4274
 *
 * Dependency-bound like every intra decoder: a TU reads reconstructed samples of earlier
 * TUs.  The host walks the list in decode order, gives each TU a wavefront level (1 + the
 * highest level among the 4x4 blocks its available neighbours lie in) and the library
 * launches one kernel per level; a wave owns one TU.  The 4n+1 neighbours live in LDS in
 * scan order (left column bottom-up, corner, top row left-to-right): in that order the
 * reference's substitution is "nearest available sample at or before me, else the first
 * available one", and its [1 2 1] smoothing is a 3-tap filter with untouched ends.
 */
#include "ffhip_internal.h"

#include <algorithm>
#include <vector>

struct HevcIntraArgs {
    const ffhip_hevc_tu *tus;
    const uint32_t *work; /* TU indices of this level */
    const int16_t *residual;
    int16_t *plane[3];
    int stride[3];
    int bitdepth_y, bitdepth_c, count;
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

__device__ static const signed char kAngle[33] = {32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                                  -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
__device__ static const short kInvAngle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256,
                                               -315, -390, -482, -630, -910, -1638, -4096};

#define NB_MAX 132 /* 4*32 + 1, padded */

__global__ __launch_bounds__(256) void k_hevc_intra(HevcIntraArgs a)
{
    __shared__ int nbA[4][NB_MAX], nbB[4][NB_MAX], refs[4][140];
    __shared__ short resl[4][32 * 32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + w;
    if (item >= a.count) return;
    const ffhip_hevc_tu t = a.tus[a.work[item]];
    const int n = 1 << t.log2_size, lg = t.log2_size, cidx = t.cidx, mode = t.pred_mode, flags = t.flags;
    const int bd = cidx == 0 ? a.bitdepth_y : a.bitdepth_c;
    int16_t *plane = a.plane[cidx];
    const int stride = a.stride[cidx];
    const int x0 = t.x, y0 = t.y, cnt = 4 * n + 1;
    int *s = nbA[w], *s2 = nbB[w], *ref = refs[w] + 34;

    /* availability in scan order: i < 2n -> left[2n-1-i]; i == 2n -> corner; i > 2n -> top[i-2n-1] */
    const unsigned long long rl = __brevll(t.avail_left) >> (64 - 2 * n); /* bit i = left[2n-1-i] */
    unsigned long long m0, m1;
    unsigned m2;
    {
        const unsigned long long c = (flags & 1) ? 1ull : 0ull;
        const unsigned long long tp = n == 32 ? t.avail_top : (t.avail_top & ((1ull << (2 * n)) - 1));
        if (n == 32) { /* left 0..63, corner 64, top 65..128 */
            m0 = rl; m1 = c | (tp << 1); m2 = (unsigned)(tp >> 63);
        } else {
            m0 = rl | (c << (2 * n)) | (tp << (2 * n + 1));
            m1 = (2 * n + 1) ? (tp >> (63 - 2 * n)) : 0; /* bits that spill past 64 (n = 16: 4n+1 = 65) */
            m2 = 0;
        }
    }
    const int n_avail = __popcll(m0) + __popcll(m1) + (int)m2;

    /* ---- 1. gather + 2. substitute ---- */
    for (int i = lane; i < cnt; i += 64) {
        int px, py;
        if (i < 2 * n) { px = x0 - 1; py = y0 + (2 * n - 1 - i); }
        else if (i == 2 * n) { px = x0 - 1; py = y0 - 1; }
        else { px = x0 + (i - 2 * n - 1); py = y0 - 1; }
        const bool av = i < 64 ? (m0 >> i) & 1 : (i < 128 ? (m1 >> (i - 64)) & 1 : m2 & 1);
        s[i] = av ? (int)plane[(long long)py * stride + px] : 0;
    }
    wave_sync();
    if (n_avail < cnt) {
        for (int i = lane; i < cnt; i += 64) {
            int v;
            if (n_avail == 0) v = 1 << (bd - 1);
            else {
                /* nearest available index <= i, else the first available one */
                int j = -1;
                if (i >= 128 && m2) j = 128;
                if (j < 0 && i >= 64) {
                    const unsigned long long mm = i >= 127 ? m1 : (m1 & ((2ull << (i - 64)) - 1));
                    if (mm) j = 127 - __clzll(mm);
                }
                if (j < 0) {
                    const unsigned long long mm = i >= 63 ? m0 : (m0 & ((2ull << i) - 1));
                    if (mm) j = 63 - __clzll(mm);
                }
                if (j < 0) j = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : 128);
                v = s[j];
            }
            s2[i] = v;
        }
        wave_sync();
        int *tmp = s; s = s2; s2 = tmp;
    }
    /* handy accessors into the scan-order array */
#define LEFT(y) s[2 * n - 1 - (y)]
#define TOP(x) s[2 * n + 1 + (x)] /* TOP(-1) is the corner */

    /* ---- 3. neighbour smoothing (8.4.4.2.3) ---- */
    if ((flags & 4) && mode != 1 && n != 4) {
        const int d26 = iabs(mode - 26), d10 = iabs(mode - 10);
        const int thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
        if ((d26 < d10 ? d26 : d10) > thr) {
            const bool bi = (flags & 8) && cidx == 0 && n == 32 &&
                            iabs(TOP(-1) + TOP(2 * n - 1) - 2 * TOP(n - 1)) < (1 << (a.bitdepth_y - 5)) &&
                            iabs(TOP(-1) + LEFT(2 * n - 1) - 2 * LEFT(n - 1)) < (1 << (a.bitdepth_y - 5));
            const int corner = TOP(-1), l63 = bi ? LEFT(63) : 0, t63 = bi ? TOP(63) : 0;
            for (int i = lane; i < cnt; i += 64) {
                int v;
                if (bi) {
                    if (i < 2 * n) { const int y = 2 * n - 1 - i; v = y == 63 ? l63 : (corner * (63 - y) + (y + 1) * l63 + 32) >> 6; }
                    else if (i == 2 * n) v = corner;
                    else { const int x = i - 2 * n - 1; v = x == 63 ? t63 : (corner * (63 - x) + (x + 1) * t63 + 32) >> 6; }
                    v = (int)(short)v;
                } else {
                    v = (i == 0 || i == cnt - 1) ? s[i] : (int)(short)((s[i - 1] + 2 * s[i] + s[i + 1] + 2) >> 2);
                }
                s2[i] = v;
            }
            wave_sync();
            int *tmp = s; s = s2; s2 = tmp;
        }
    }

    /* ---- residual (with the optional rdpcm accumulation of 8.6.5) ---- */
    short *R = resl[w];
    const bool has_res = (flags & 2) != 0;
    if (has_res) {
        const int16_t *src = a.residual + t.res_offset;
        for (int i = lane; i < n * n; i += 64) R[i] = src[i];
        wave_sync();
        if (flags & 0x40) {
            if (mode / 26 == 0) { /* running sum over the flattened block from index n (hevc.c:3963-3968) */
                if (lane == 0)
                    for (int i = n; i < n * n; i++) R[i] = (short)(R[i] + R[i - 1]);
            } else if (lane < n) {
                for (int y = 1; y < n; y++) R[lane + n * y] = (short)(R[lane + n * y] + R[lane + n * (y - 1)]);
            }
            wave_sync();
        }
        if (flags & 0x80) { /* 8.6.6 with rY aliased to r, as at hevc.c:4753-4755; products wrap like -fwrapv */
            const int bdc = a.bitdepth_c, bdy = a.bitdepth_y;
            for (int i = lane; i < n * n; i += 64) {
                const int up = (int)((unsigned)(int)R[i] << bdc) >> bdy;
                R[i] = (short)(R[i] + ((int)((unsigned)t.res_scale * (unsigned)up) >> 3));
            }
            wave_sync();
        }
    }

    /* ---- 4. prediction (the reference reads the neighbours as uint16_t) ---- */
#define U16(v) ((int)((unsigned)(v) & 0xffffu))
    int dc = 0;
    int angle = 0;
    if (mode == 1) {
        unsigned sum = 0;
        for (int i = 0; i < n; i++) sum += (unsigned)U16(LEFT(i)) + (unsigned)U16(TOP(i));
        dc = (int)((sum + (1u << lg)) >> (lg + 1));
    } else if (mode >= 2) {
        angle = kAngle[mode - 2];
        /* ref[] of 8.4.4.2.6: main = top for modes >= 18, left otherwise; both start at the corner */
        for (int xx = lane; xx <= 2 * n; xx += 64) {
            if (xx == 0) ref[0] = U16(TOP(-1));
            else if (xx <= n || angle >= 0) ref[xx] = mode >= 18 ? U16(TOP(xx - 1)) : U16(LEFT(xx - 1));
        }
        if (angle < 0 && ((n * angle) >> 5) < -1) {
            const int lo = (angle * n) >> 5, inv = kInvAngle[mode - 11];
            for (int xx = -1 - lane; xx >= lo; xx -= 64) {
                const int k = (xx * inv + 128) >> 8;
                ref[xx] = k == 0 ? U16(TOP(-1)) : (mode >= 18 ? U16(LEFT(k - 1)) : U16(TOP(k - 1)));
            }
        }
        wave_sync();
    }
    const bool edge_ok = cidx == 0 && n < 32;
    for (int p = lane; p < n * n; p += 64) {
        const int x = p & (n - 1), y = p >> lg;
        int v;
        if (mode == 0) {
            v = ((n - 1 - x) * U16(LEFT(y)) + (x + 1) * U16(TOP(n)) + (n - 1 - y) * U16(TOP(x)) + (y + 1) * U16(LEFT(n)) + n) >> (lg + 1);
        } else if (mode == 1) {
            v = dc;
            if (edge_ok && !(flags & 0x20)) {
                if (x == 0 && y == 0) v = (U16(LEFT(0)) + 2 * dc + U16(TOP(0)) + 2) >> 2;
                else if (y == 0) v = (U16(TOP(x)) + 3 * dc + 2) >> 2;
                else if (x == 0) v = (U16(LEFT(y)) + 3 * dc + 2) >> 2;
            }
        } else {
            const int al = mode >= 18 ? y : x, ac = mode >= 18 ? x : y; /* along / across the direction */
            const int idx = ((al + 1) * angle) >> 5, fact = ((al + 1) * angle) & 31;
            v = fact ? ((32 - fact) * ref[ac + idx + 1] + fact * ref[ac + idx + 2] + 16) >> 5 : ref[ac + idx + 1];
            if (edge_ok && !(flags & 0x10)) {
                if (mode == 26 && x == 0) v = clip3i(0, (1 << a.bitdepth_y) - 1, U16(TOP(0)) + ((U16(LEFT(y)) - U16(TOP(-1))) >> 1));
                if (mode == 10 && y == 0) v = clip3i(0, (1 << a.bitdepth_y) - 1, U16(LEFT(0)) + ((U16(TOP(x)) - U16(TOP(-1))) >> 1));
            }
        }
        /* ---- 5. reconstruct: pred is stored as int16 by the reference before the add ---- */
        const int pr = (int)(short)(v & 0xffff);
        const int rs = has_res ? (int)R[p] : 0;
        plane[(long long)(y0 + y) * stride + x0 + x] = (short)clip3i(0, (1 << bd) - 1, pr + rs);
    }
}

/* ------------------------------------------------------------------------ host */

static uint32_t *g_work = nullptr;
static size_t g_work_cap = 0;

extern "C" int ffhip_hevc_intra_recon(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus,
                                      const int16_t *d_residual, int16_t *d_y, int16_t *d_cb, int16_t *d_cr,
                                      int width_y, int height_y, int y_stride, int width_c, int height_c,
                                      int uv_stride, int bitdepth_y, int bitdepth_c, void *stream)
{
    if (n_tus < 0 || n_tus > 0x7fffffffLL) return FFHIP_EINVAL;
    if (n_tus == 0) return FFHIP_OK;
    if (!h_tus || !d_tus || !d_y || width_y <= 0 || height_y <= 0 || y_stride < width_y) return FFHIP_EINVAL;
    if (bitdepth_y < 8 || bitdepth_y > 15 || bitdepth_c < 8 || bitdepth_c > 15) return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    const int pw[3] = {width_y, width_c, width_c}, ph[3] = {height_y, height_c, height_c};
    /* wavefront levels at 4x4-block granularity, per plane */
    std::vector<int> lvl[3];
    int bw[3];
    for (int c = 0; c < 3; c++) {
        bw[c] = (pw[c] + 3) / 4;
        lvl[c].assign((size_t)(c == 0 || (d_cb && d_cr) ? bw[c] * ((ph[c] + 3) / 4) : 0), -1);
    }
    std::vector<std::vector<uint32_t>> lists;
    bool has_res = false;
    for (long long i = 0; i < n_tus; i++) {
        const ffhip_hevc_tu &t = h_tus[i];
        const int c = t.cidx, n = 1 << t.log2_size;
        if (c > 2 || t.log2_size < 2 || t.log2_size > 5 || t.pred_mode > 34) return FFHIP_EINVAL;
        if (c > 0 && (!d_cb || !d_cr || uv_stride < width_c)) return FFHIP_EINVAL;
        if (t.x + n > pw[c] || t.y + n > ph[c]) return FFHIP_EINVAL;
        int lv = 0;
        auto dep = [&](int px, int py) {
            if (px < 0 || py < 0 || px >= pw[c] || py >= ph[c]) return false; /* a mask bit outside the plane */
            lv = std::max(lv, lvl[c][(size_t)(py / 4) * bw[c] + px / 4] + 1);
            return true;
        };
        if ((t.flags & 1) && !dep(t.x - 1, t.y - 1)) return FFHIP_EINVAL;
        for (int k = 0; k < 2 * n; k++) {
            if (((t.avail_top >> k) & 1) && !dep(t.x + k, t.y - 1)) return FFHIP_EINVAL;
            if (((t.avail_left >> k) & 1) && !dep(t.x - 1, t.y + k)) return FFHIP_EINVAL;
        }
        for (int by = t.y / 4; by < (t.y + n) / 4; by++)
            for (int bx = t.x / 4; bx < (t.x + n) / 4; bx++) lvl[c][(size_t)by * bw[c] + bx] = lv;
        if ((size_t)lv >= lists.size()) lists.resize((size_t)lv + 1);
        lists[(size_t)lv].push_back((uint32_t)i);
        has_res |= (t.flags & 2) != 0;
    }
    if (has_res && !d_residual) return FFHIP_EINVAL;
    if ((size_t)n_tus > g_work_cap) {
        if (g_work) (void)hipFree(g_work);
        g_work = nullptr;
        g_work_cap = 0;
        FFHIP_CHECK(hipMalloc((void **)&g_work, (size_t)n_tus * sizeof(uint32_t)), FFHIP_ENOMEM);
        g_work_cap = (size_t)n_tus;
    }
    std::vector<uint32_t> flat;
    flat.reserve((size_t)n_tus);
    for (auto &l : lists) flat.insert(flat.end(), l.begin(), l.end());
    hipStream_t st = (hipStream_t)stream;
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
    FFHIP_CHECK(hipMemcpy(g_work, flat.data(), flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice), FFHIP_EIO);
    HevcIntraArgs a;
    a.tus = d_tus; a.residual = d_residual;
    a.plane[0] = d_y; a.plane[1] = d_cb; a.plane[2] = d_cr;
    a.stride[0] = y_stride; a.stride[1] = uv_stride; a.stride[2] = uv_stride;
    a.bitdepth_y = bitdepth_y; a.bitdepth_c = bitdepth_c;
    size_t off = 0;
    for (auto &l : lists) {
        a.work = g_work + off;
        a.count = (int)l.size();
        hipLaunchKernelGGL(k_hevc_intra, dim3((unsigned)((a.count + 3) / 4)), dim3(256), 0, st, a);
        off += l.size();
    }
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
