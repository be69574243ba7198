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

#include <algorithm>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct Vp8LfArgs {
    const uint8_t *modes;   /* [n_images][n_mb][20]: [0] intra_y_mode, [18] segment_id */
    const uint8_t *filters; /* [4][2][3] sub_limit, inter_limit, hev_thresh             */
    const uint32_t *work;   /* (image, mb) pairs of this level                          */
    uint8_t *y, *u, *v;
    long long plane_y, plane_uv;
    int mbcols, mbrows, count, filter_type;
    /* row form only */
    int pshift, pred_pshift; /* a row's progress counter is word (image * mbrows + row) << pshift; the prediction's likewise */
    uint32_t *ctrl; /* [0] next row ticket, [FFHIP_VP8_LF_CTRL_ABORT] abort; from ctrl + FFHIP_VP8_LF_CTRL_HDR: macroblocks finished per (image, row) */
    int *async_err;
    int n_images;
    int slack; /* as in Vp8PredArgs */
    int debug_giveup;              /* test hook: a fused launch gives up by itself (the self-healing path of ffhip_stream_sync) */
    const uint32_t *pred_progress; /* fused with the prediction (ffhip_vp8_predict_loopfilter): its per-(image, row) counters, else null */
    int pred_split;                /* that prediction runs its chroma as rows of their own: their counters follow the luma rows' */
};

#include "ffhip_vp8_filters.h"


#define LS 24 /* luma tile row stride (4 + 16, padded)   */
#define CS 24 /* chroma tile row stride (4 + 8, padded to the luma stride: the row kernel walks luma and chroma columns with the same offsets) */

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

/* Row form: ONE launch per batch, the scheme of k_vp8_predict_rows.  A wave owns one macroblock row
 * of one image; macroblock (x, y) may start once (x + 1, y - 1) is out, which the row above says
 * through its progress counter.  Going left to right, the 4-pixel left border is simply the right
 * end of the tile the wave has just filtered, so per macroblock only the new 16 (8) columns are
 * fetched -- one macroblock ahead, with device-coherent loads because the four rows above belong
 * to another wave -- and the filtered columns are stored with agent-scope stores.  The counter moves
 * when the in-order completion of the wave's memory operations proves those stores done. */
#define LF_SPIN_LIMIT (1 << 21)

/* per lane: bit `lane` of a 64-bit mask picks b over a -- one v_cndmask with the mask in an SGPR pair */
__device__ __forceinline__ int lane_select(unsigned long long mask, int a, int b)
{
    int d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(mask));
    return d;
}
struct LfFetch {
    u32 y0, y1, cu, cv; /* luma items lane and lane + 64 (of 80: 20 rows x 4 dwords), one chroma dword of U and of V (of 24 each: 12 rows x 2) */
    u32 m0, m4;         /* mode bytes 0..3 and 16..19 of the macroblock */
};

/* per lane, by its number alone: everything the macroblock loop would otherwise work out per macroblock (divisions by 5
 * and 3, row / column tests, a choice of plane per lane that the compiler turns into a loop over the planes) */
#define LF_OUT ((int)0x80000000u) /* an offset outside every plane: loads return 0, stores are dropped (buffer range check) */

template <int TYPE> /* the filter type of the launch (1 simple, 2 normal): a kernel each, no run-time switch in the edges */
__global__ __launch_bounds__(64) void k_vp8_loopfilter_rows(Vp8LfArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t TL[20 * LS];
    __shared__ __attribute__((aligned(16))) uint8_t TC[2][20 * CS]; /* 12 rows in use; 20 so that filter_phase's unused reads of a chroma column stay inside */
    __shared__ uint8_t FT[24];
    __shared__ u32 DUMP[64]; /* where lanes without a role read and write */
    static_assert(LS == CS, "filter_phase walks luma and chroma tiles with one stride");
    const int lane = threadIdx.x;
    const int n_mb = a.mbcols * a.mbrows;
    const int ys = 16 * a.mbcols, us = 8 * a.mbcols;
    constexpr int type = TYPE;
    uint32_t *progress = a.ctrl + FFHIP_VP8_LF_CTRL_HDR;
    if (lane < 24) FT[lane] = a.filters[lane]; /* a read from memory per macroblock would drain the fetches in flight */
    wave_sync();
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    const unsigned tl = (unsigned)(unsigned long long)(lds_u8 *)TL, tc0 = (unsigned)(unsigned long long)(lds_u8 *)TC[0], tc1 = (unsigned)(unsigned long long)(lds_u8 *)TC[1],
                   dump = (unsigned)(unsigned long long)(lds_u8 *)(uint8_t *)DUMP + 4u * (unsigned)lane;
#define LDS32(addr) (*(__attribute__((address_space(3))) u32 *)(unsigned long long)(addr)) /* a dword of LDS at a byte address kept in a register */
    /* fetch: plane offsets of the lane's items relative to the macroblock's first pixel, and where they go in the tiles */
    const int fy0 = ((lane >> 2) - 4) * ys + 4 * (lane & 3);                       /* item lane: tile row lane >> 2 (picture row - 4) */
    const int fy1 = lane < 16 ? ((lane >> 2) + 12) * ys + 4 * (lane & 3) : LF_OUT; /* item lane + 64                                  */
    const int fc = lane < 24 ? ((lane >> 1) - 4) * us + 4 * (lane & 1) : LF_OUT;   /* chroma item lane of either plane                */
    const unsigned dy0 = tl + (unsigned)((lane >> 2) * LS + 4 + 4 * (lane & 3));
    const unsigned dy1 = lane < 16 ? tl + (unsigned)(((lane >> 2) + 16) * LS + 4 + 4 * (lane & 3)) : dump;
    const unsigned dcu = lane < 24 ? tc0 + (unsigned)((lane >> 1) * CS + 4 + 4 * (lane & 1)) : dump;
    const unsigned dcv = lane < 24 ? tc1 + (unsigned)((lane >> 1) * CS + 4 + 4 * (lane & 1)) : dump;
    /* the tile's right end becomes the next macroblock's left border: lanes 0-19 a luma row, 32-43 / 44-55 a chroma row */
    const unsigned keep_src = lane < 20 ? tl + (unsigned)(lane * LS + 16) : (lane >= 32 && lane < 56 && type != 1 ? (lane < 44 ? tc0 : tc1) + (unsigned)(((lane - 32) % 12) * CS + 8) : dump);
    const unsigned keep_dst = lane < 20 ? tl + (unsigned)(lane * LS) : (lane >= 32 && lane < 56 && type != 1 ? (lane < 44 ? tc0 : tc1) + (unsigned)(((lane - 32) % 12) * CS) : dump);
    /* write-back: luma items lane and lane + 64 of 100 (20 rows x 5 dwords, the first one the 4 columns left of the
     * macroblock), chroma item lane of 36 (12 rows x 3) of either plane.  An item in tile rows 0-3 exists only below the
     * first macroblock row, one in dword column 0 only right of the first macroblock: as lane masks, picked per macroblock */
    const int wr0 = lane / 5, wd0 = lane % 5, wr1 = (lane + 64) / 5, wd1 = (lane + 64) % 5, wrc = lane / 3, wdc = lane % 3;
    const int wy0 = (wr0 - 4) * ys + 4 * wd0 - 4, wy1 = lane < 36 ? (wr1 - 4) * ys + 4 * wd1 - 4 : LF_OUT, wc = lane < 36 ? (wrc - 4) * us + 4 * wdc - 4 : LF_OUT;
    const unsigned sy0 = tl + (unsigned)(wr0 * LS + 4 * wd0), sy1 = lane < 36 ? tl + (unsigned)(wr1 * LS + 4 * wd1) : dump;
    const unsigned scu = lane < 36 ? tc0 + (unsigned)(wrc * CS + 4 * wdc) : dump, scv = lane < 36 ? tc1 + (unsigned)(wrc * CS + 4 * wdc) : dump;

    for (;;) {
        unsigned ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(&a.ctrl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
        if (ticket >= (unsigned)(a.n_images * a.mbrows)) return;
        const int y = (int)(ticket / (unsigned)a.n_images), img = (int)(ticket % (unsigned)a.n_images);
        if (a.debug_giveup && a.pred_progress && y == a.mbrows / 2) { /* test hook (FFHIP_DEBUG_VP8_LF_GIVEUP): as if a wait for the prediction had run out half-way down */
            if (lane == 0) {
                __hip_atomic_store(&a.ctrl[FFHIP_VP8_LF_CTRL_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.async_err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        uint8_t *Y = a.y + (long long)img * a.plane_y;
        uint8_t *P[2] = {a.u + (long long)img * a.plane_uv, a.v + (long long)img * a.plane_uv};
        const uint8_t *mrow = a.modes + ((long long)img * n_mb + (long long)y * a.mbcols) * 20;
        const uint32_t *prog_up = progress + (((long long)img * a.mbrows + y - 1) << a.pshift);
        uint32_t *prog_me = progress + (((long long)img * a.mbrows + y) << a.pshift);
        unsigned seen = y == 0 ? 0x7fffffffu : 0u;
        /* Running NEXT TO the prediction kernel (another stream of the same call): macroblock (x, y) may be filtered once the
         * prediction has finished (x + 1, y + 1) -- the filter rewrites row 15 and columns 13-15 of what the prediction of the
         * row below and of the right neighbour still reads unfiltered (predict.c reads reconstructed, not filtered, samples),
         * and rows 13-15 of the row above, which the prediction of this row read.  The last row has no row below. */
        const uint32_t *pred_row = a.pred_progress ? a.pred_progress + (((long long)img * a.mbrows + (y + 1 < a.mbrows ? y + 1 : y)) << a.pred_pshift) : nullptr;
        unsigned seen_pred = a.pred_progress ? 0u : 0x7fffffffu, seen_pred_c = seen_pred;
        const uint32_t *pred_row_c = pred_row ? pred_row + (((long long)a.n_images * a.mbrows) << a.pred_pshift) : nullptr; /* the chroma rows' counters follow the luma rows' */
        const __amdgpu_buffer_rsrc_t rY = ffhip_rsrc(Y, 256u * (unsigned)n_mb), rU = ffhip_rsrc(P[0], 64u * (unsigned)n_mb),
                                     rV = ffhip_rsrc(P[1], 64u * (unsigned)n_mb);
        const int row_org = y * 16 * ys, row_corg = y * 8 * us;
        /* lanes whose write-back items exist at all in this row (tile rows 0-3 need a row above), and those of them that also
         * exist in the first macroblock (dword column 0 needs a macroblock to the left) */
        const unsigned long long ok0 = __builtin_amdgcn_ballot_w64(wr0 >= 4 || y > 0), ok1 = __builtin_amdgcn_ballot_w64(wr1 >= 4 || y > 0),
                                 okc = __builtin_amdgcn_ballot_w64(wrc >= 4 || y > 0);
        const unsigned long long first0 = ok0 & __builtin_amdgcn_ballot_w64(wd0 > 0), first1 = ok1 & __builtin_amdgcn_ballot_w64(wd1 > 0),
                                 firstc = okc & __builtin_amdgcn_ballot_w64(wdc > 0);

        /* one poll of the row above always in flight: issued behind a fetch, read in front of the next (k_vp8_predict_rows) */
        u32 poll_v = 0;
        bool poll_out = false;
        auto fetch = [&](int x1, LfFetch &f) -> bool {
            const unsigned need = y == 0 ? 0u : (unsigned)(x1 + 2 < a.mbcols ? x1 + 2 : a.mbcols);
            int spins = 0;
            if (poll_out && seen < need) {
                const unsigned got = (unsigned)__builtin_amdgcn_readfirstlane((int)poll_v);
                seen = got > seen ? got : seen;
            }
            poll_out = false;
            const unsigned want = seen < need ? ((need + (unsigned)a.slack) < (unsigned)a.mbcols ? need + (unsigned)a.slack : (unsigned)a.mbcols) : need;
            while (seen < want) {
                seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(prog_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen >= want) break;
                if ((spins & 15) == 15 && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&a.ctrl[FFHIP_VP8_LF_CTRL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return false; /* given up elsewhere: whoever did has said why */
                if (++spins > LF_SPIN_LIMIT) {
                    if (lane == 0) {
                        __hip_atomic_store(&a.ctrl[FFHIP_VP8_LF_CTRL_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(a.async_err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    return false;
                }
                if (spins < 16) __builtin_amdgcn_s_sleep(1);
                else __builtin_amdgcn_s_sleep(16);
            }
            const unsigned need_pred = (unsigned)(x1 + 2 < a.mbcols ? x1 + 2 : a.mbcols);
            /* the prediction's luma rows and (normal filter: it rewrites chroma too) its chroma rows, which are rows of their own
             * with counters of their own since round 3 and run ahead of the luma: their counter is polled only while it is short */
            auto wait_pred = [&](const uint32_t *row, unsigned &seen_p) -> bool {
                const unsigned want_pred = seen_p < need_pred ? ((need_pred + (unsigned)a.slack) < (unsigned)a.mbcols ? need_pred + (unsigned)a.slack : (unsigned)a.mbcols) : need_pred;
                while (seen_p < want_pred) {
                    seen_p = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (seen_p >= want_pred) break;
                    /* the prediction was refused (bad mode bytes) or gave up: it has reported why; this kernel just stops (the prediction's
                     * abort word sits three words in front of its counters) */
                    /* looked at every 64th poll only: the word shares its cache line with the prediction's ticket counter, and a load per
                     * poll from every waiting filter wave slowed a 1024-frame call by a third */
                    if ((spins & 63) == 63 && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(a.pred_progress - FFHIP_VP8_CTRL_HDR + FFHIP_VP8_CTRL_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        if (lane == 0) __hip_atomic_store(&a.ctrl[FFHIP_VP8_LF_CTRL_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return false;
                    }
                    if ((spins & 15) == 15 && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&a.ctrl[FFHIP_VP8_LF_CTRL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return false; /* given up elsewhere: whoever did has said why */
                    if (++spins > LF_SPIN_LIMIT) {
                        if (lane == 0) {
                            __hip_atomic_store(&a.ctrl[FFHIP_VP8_LF_CTRL_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(a.async_err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        return false;
                    }
                    if (spins < 16) __builtin_amdgcn_s_sleep(1);
                    else __builtin_amdgcn_s_sleep(16);
                }
                return true;
            };
            if (!wait_pred(pred_row, seen_pred)) return false;
            if (type != 1 && a.pred_split && !wait_pred(pred_row_c, seen_pred_c)) return false;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); /* ordering only */
            const u32 *mp = (const u32 *)(mrow + (long long)x1 * 20);
            f.m0 = mp[0];
            f.m4 = mp[4];
            /* unconditional loads, device-coherent (the four rows above belong to another wave): a lane without an item, and
             * an item above the picture, read as 0 by the buffer range check and are never used */
            const int org = row_org + x1 * 16, corg = row_corg + x1 * 8;
            f.y0 = (u32)__builtin_amdgcn_raw_buffer_load_b32(rY, fy0 + org, 0, FFHIP_AUX_SC1);
            f.y1 = (u32)__builtin_amdgcn_raw_buffer_load_b32(rY, fy1 + org, 0, FFHIP_AUX_SC1);
            if (type != 1) {
                f.cu = (u32)__builtin_amdgcn_raw_buffer_load_b32(rU, fc + corg, 0, FFHIP_AUX_SC1);
                f.cv = (u32)__builtin_amdgcn_raw_buffer_load_b32(rV, fc + corg, 0, FFHIP_AUX_SC1);
            }
            if (seen < (unsigned)a.mbcols) {
                poll_v = __hip_atomic_load(prog_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                poll_out = true;
            }
            return true;
        };

        /* fetches run ONE macroblock ahead, into the registers the consumed fetch has just left (two ahead through register
         * sets rotated by copies, as in the first form, made every macroblock wait for the loads it had just issued: a copy of
         * a register with a load in flight waits for the load) */
        LfFetch f;
        f.cu = f.cv = 0;
        if (!fetch(0, f)) return;
        /* The fetch of macroblock x + 1 is consumed at the END of macroblock x, in front of x's stores (k_vp8_predict_rows has the
         * reasoning: consumed at the loop's top, the wait for the fetched registers was a wait for the stores just issued to
         * complete).  The tile shifts when the fetch goes in, so the four dwords a lane writes back are read from the tile first. */
        u32 m0 = 0, m4 = 0; /* mode bytes 0..3 and 16..19 of the macroblock at hand */
        auto consume = [&]() {
            /* the tile's right end becomes the left border, the fetch the new columns */
            const u32 keep = LDS32(keep_src);
            wave_sync();
            LDS32(keep_dst) = keep;
            LDS32(dy0) = f.y0;
            LDS32(dy1) = f.y1;
            if (type != 1) {
                LDS32(dcu) = f.cu;
                LDS32(dcv) = f.cv;
            }
            m0 = (u32)__builtin_amdgcn_readfirstlane((int)f.m0);
            m4 = (u32)__builtin_amdgcn_readfirstlane((int)f.m4);
            wave_sync();
        };
        consume();
        for (int x = 0; x < a.mbcols; x++) {
            /* the fetch consumed at the end of the previous iteration was issued behind the stores of macroblock x - 2: those are done */
            if (lane == 0 && x >= 2) __hip_atomic_store(prog_me, (unsigned)(x - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x + 1 < a.mbcols && !fetch(x + 1, f)) return;

            const bool bpred = (m0 & 0xff) == 4;
            const uint8_t *fp = FT + ((((m4 >> 16) & 3) * 2) + (bpred ? 1 : 0)) * 3;
            const int sub = __builtin_amdgcn_readfirstlane((int)fp[0]), inter = __builtin_amdgcn_readfirstlane((int)fp[1]),
                      hevt = __builtin_amdgcn_readfirstlane((int)fp[2]);
            u32 wb0 = 0, wb1 = 0, wbu = 0, wbv = 0;
            if (sub) { /* wave-uniform */
                /* webp.c:1710-1745: inner edges for B_PRED MBs in the simple filter, for the others in the normal one */
                const bool inner = type == 1 ? bpred : !bpred;
                /* ---- vertical edges: one pixel row per lane (lanes 0-15 luma, 16-23 U, 24-31 V), then horizontal edges: one
                 * pixel column per lane ---- */
                const bool lum = lane < 16, active = lane < 16 || (lane < 32 && type != 1);
                uint8_t *const mine = lum ? TL : TC[(lane >> 3) & 1];
                const int li = lum ? lane : (lane & 7);
                filter_phase<1, TYPE>(mine + (li + 4) * LS, active, lum, x > 0, inner, sub, inter, hevt);
                wave_sync();
                filter_phase<LS, TYPE>(mine + li + 4, active, lum, y > 0, inner, sub, inter, hevt);
                wave_sync();
                wb0 = LDS32(sy0); wb1 = LDS32(sy1);
                if (type != 1) { wbu = LDS32(scu); wbv = LDS32(scv); }
            }
            wave_sync();
            if (x + 1 < a.mbcols) consume();
            if (sub) {
                /* ---- write back rows -4..15, columns -4..15 as dwords (the cells this macroblock did not change are
                 * rewritten with the value it read: their owners are finished) -- but nothing outside the picture ---- */
                const int org = row_org + x * 16, corg = row_corg + x * 8;
                __builtin_amdgcn_raw_buffer_store_b32(wb0, rY, lane_select(x > 0 ? ok0 : first0, LF_OUT, wy0) + org, 0, FFHIP_AUX_SC1);
                __builtin_amdgcn_raw_buffer_store_b32(wb1, rY, lane_select(x > 0 ? ok1 : first1, LF_OUT, wy1) + org, 0, FFHIP_AUX_SC1);
                if (type != 1) {
                    const int oc = lane_select(x > 0 ? okc : firstc, LF_OUT, wc) + corg; /* added here, not as the scalar offset: the range check looks at this operand alone, and a lane's offset may be negative */
                    __builtin_amdgcn_raw_buffer_store_b32(wbu, rU, oc, 0, FFHIP_AUX_SC1);
                    __builtin_amdgcn_raw_buffer_store_b32(wbv, rV, oc, 0, FFHIP_AUX_SC1);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(prog_me, (unsigned)a.mbcols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define SCRATCH_VP8_LF 2

thread_local FfhipVp8Fusion g_ffhip_vp8_fusion = {0, nullptr, nullptr, nullptr, 0};
extern "C" void ffhip_vp8_note_enqueue(void *stream); /* below: the self-healing record of a side-by-side call is good while that call is the stream's last */

extern "C" int ffhip_vp8_loopfilter(int mbcols, int mbrows, int n_images, int filter_type, const uint8_t *d_modes,
                                    const uint8_t *d_filters, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                                    int64_t plane_stride_y, int64_t plane_stride_uv, void *stream)
{
    if (mbcols <= 0 || mbrows <= 0 || n_images < 0 || filter_type < 0 || filter_type > 2) return FFHIP_EINVAL;
    if (n_images == 0 || filter_type == 0) return FFHIP_OK; /* WEBP_FILTER_NONE (webp.c:1852-1856) */
    if (!d_modes || !d_filters || !d_y || !d_u || !d_v) return FFHIP_EINVAL;
    ffhip_vp8_note_enqueue(stream);
    const long long n_mb = (long long)mbcols * mbrows;
    if (n_mb * n_images > 0x3fffffffLL) return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    hipStream_t st = (hipStream_t)stream;
    /* row form (default): one launch, no host-side scheduling */
    const char *mode_env = FFHIP_ENV("FFHIP_VP8_LF_MODE");
    int *async_err = (mode_env && !strcmp(mode_env, "levels")) ? nullptr : ffhip_async_err_word();
    if (async_err && g_ffhip_vp8_fusion.active && g_ffhip_vp8_fusion.err_word) async_err = g_ffhip_vp8_fusion.err_word; /* the side-by-side call's own word */
    if (async_err && !((uintptr_t)d_modes & 3) && !(((uintptr_t)d_y | (uintptr_t)d_u | (uintptr_t)d_v | (uintptr_t)plane_stride_y | (uintptr_t)plane_stride_uv) & 3) &&
        n_mb < (1LL << 23)) {
        const int pshift = []{ const char *e = FFHIP_ENV("FFHIP_VP8_PROGRESS_SHIFT"); return e ? std::min(5, std::max(0, atoi(e))) : 5; }();
        const size_t words = FFHIP_VP8_LF_CTRL_HDR + (((size_t)n_images * (size_t)mbrows) << pshift);
        uint32_t *g_work = ffhip_scratch(SCRATCH_VP8_LF, stream, words);
        if (!g_work) return FFHIP_ENOMEM;
        const uint32_t *pred_progress = nullptr;
        if (g_ffhip_vp8_fusion.active && g_ffhip_vp8_fusion.pred_progress) { /* next to the prediction kernel, behind its counter reset */
            pred_progress = g_ffhip_vp8_fusion.pred_progress;
            st = (hipStream_t)g_ffhip_vp8_fusion.side;
            FFHIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)g_ffhip_vp8_fusion.fork, 0), FFHIP_EIO);
        }
        FFHIP_CHECK(hipMemsetAsync(g_work, 0, words * sizeof(uint32_t), st), FFHIP_EIO);
        Vp8LfArgs a = {};
        a.pred_progress = pred_progress; a.pshift = pshift; a.pred_pshift = g_ffhip_vp8_fusion.pshift;
        a.pred_split = pred_progress ? g_ffhip_vp8_fusion.pred_split : 0;
        a.modes = d_modes; a.filters = d_filters; a.y = d_y; a.u = d_u; a.v = d_v;
        a.plane_y = plane_stride_y; a.plane_uv = plane_stride_uv;
        a.mbcols = mbcols; a.mbrows = mbrows; a.filter_type = filter_type;
        a.ctrl = g_work; a.async_err = async_err; a.n_images = n_images;
        { const char *sl = FFHIP_ENV("FFHIP_VP8_SLACK"); a.slack = sl ? std::max(0, atoi(sl)) : 0; }
        a.debug_giveup = FFHIP_ENV("FFHIP_DEBUG_VP8_LF_GIVEUP") ? 1 : 0;
        /* as many waves as can be resident, at most a wavefront's width of rows per image (ffhip_vp8_predict_recon has the
         * reasoning); next to the prediction kernel of the same call each of the two takes half of its own residency, so
         * filter waves -- which wait for the prediction's counters -- can never keep the prediction from becoming resident */
        const char *wv = FFHIP_ENV("FFHIP_VP8_LF_WAVES");
        /* measured (256 x 1080p): the filter is at its best with two waves per SIMD and loses a factor of two with four (1024 / 2048 /
         * 3584 waves: 1.95 / 1.86 / 3.4 ms -- its waves mostly wait, and waiting waves poll); next to the prediction one per SIMD does */
        long long resident = ffhip_resident_waves(filter_type == 1 ? (const void *)k_vp8_loopfilter_rows<1> : (const void *)k_vp8_loopfilter_rows<2>, 64);
        resident = std::max<long long>(1, pred_progress ? resident / 4 : resident / 3);
        const long long wide = std::max<long long>(256, (long long)n_images * (mbcols / 8 + 1)); /* measured, 16 x 1080p: 8 / 16 / 24 / 32 rows per image 0.83 / 0.67 / 0.81 / 0.85 ms alone (encoder's stream) */
        const long long cap = wv ? std::max(1, atoi(wv)) : std::min(resident, wide);
        const dim3 grid((unsigned)std::min<long long>((long long)n_images * mbrows, cap));
        if (filter_type == 1) hipLaunchKernelGGL(k_vp8_loopfilter_rows<1>, grid, dim3(64), 0, st, a);
        else hipLaunchKernelGGL(k_vp8_loopfilter_rows<2>, grid, dim3(64), 0, st, a);
        FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
        return FFHIP_OK;
    }

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
    uint32_t *g_work = ffhip_scratch(SCRATCH_VP8_LF, stream, total);
    if (!g_work) return FFHIP_ENOMEM;
    std::vector<uint32_t> flat;
    flat.reserve(total);
    for (auto &l : lists) flat.insert(flat.end(), l.begin(), l.end());
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
    FFHIP_CHECK(hipMemcpy(g_work, flat.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice), FFHIP_EIO);
    Vp8LfArgs a = {};
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

/* calculate_filter_control_parameter (format/webp.c:1756-1803) for the four segments and both macroblock kinds, as
 * WEBP_read_frame calls it (webp.c:1905-1915: once per DCT partition index, not per segment).  Host arithmetic, no device.  A level of 0 only clears sub_limit
 * (webp.c:1799); the other two fields keep what the decoder's zero-initialised state held. */
extern "C" int ffhip_vp8_filter_params(const ffhip_vp8_filter_header *h, uint8_t *filters, int *filter_type)
{
    if (!h || !filters || !filter_type || h->loop_filter_level > 63 || h->sharpness_level > 7 ||
        (h->nbr_partitions != 1 && h->nbr_partitions != 2 && h->nbr_partitions != 4 && h->nbr_partitions != 8)) return FFHIP_EINVAL;
    const int ft = h->loop_filter_level == 0 ? 0 : (h->filter_type ? 1 : 2);
    *filter_type = ft;
    for (int i = 0; i < 24; i++) filters[i] = 0;
    if (!ft) return FFHIP_OK;
    auto clamp63 = [](int v) { return v < 0 ? 0 : (v > 63 ? 63 : v); };
    for (int s = 0; s < 4 && s < h->nbr_partitions; s++)
        for (int is4 = 0; is4 < 2; is4++) {
            int base = h->loop_filter_level;
            if (h->segmentation_enabled) base = h->segment_feature_mode ? h->lf_update_value[s] : base + h->lf_update_value[s];
            int level = clamp63(base);
            if (h->loop_filter_adj_enable) level += h->mode_ref_lf_delta0 + (is4 ? h->mb_mode_delta0 : 0);
            level = clamp63(level);
            uint8_t *f = filters + (s * 2 + is4) * 3;
            if (level > 0) {
                int il = level;
                if (h->sharpness_level > 0) {
                    il >>= h->sharpness_level > 4 ? 2 : 1;
                    if (il > 9 - h->sharpness_level) il = 9 - h->sharpness_level;
                }
                if (il < 1) il = 1;
                f[0] = (uint8_t)((level << 1) + il);
                f[1] = (uint8_t)il;
                f[2] = level >= 40 ? 2 : (level >= 15 ? 1 : 0);
            }
        }
    return FFHIP_OK;
}

/* The side stream and the two events of ffhip_vp8_predict_loopfilter: one set per calling thread and device, made on first
 * use, recreated when the thread's current device has changed, released by ffhip_shutdown (ffhip_vp8_release_side_streams)
 * or when the thread ends. */
namespace {
struct SideStream {
    int device = -1;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, mid = nullptr, aux = nullptr;
    /* ffhip_hevc_intra_recon_tiles' pipeline (made on first use): a stream for the chunks' pre-passes, a second stream for grouped kernels, events */
    hipStream_t plan = nullptr, groups2 = nullptr;
    hipEvent_t pev[FFHIP_PIPE_EVENTS] = {};
    /* ffhip_jpeg_entropy_batch_gpu's (made on first use, all or none) */
    hipStream_t huff_up = nullptr, huff_c2 = nullptr;
    hipEvent_t huff_ev[FFHIP_HUFF_PARTS + 4] = {};
};
std::mutex g_side_mu;
std::vector<SideStream *> g_sides;
void side_release(SideStream *s)
{
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->fork) (void)hipEventDestroy(s->fork);
    if (s->join) (void)hipEventDestroy(s->join);
    if (s->mid) (void)hipEventDestroy(s->mid);
    if (s->aux) (void)hipEventDestroy(s->aux);
    if (s->plan) (void)hipStreamDestroy(s->plan);
    if (s->groups2) (void)hipStreamDestroy(s->groups2);
    for (auto &e : s->pev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (s->huff_up) (void)hipStreamDestroy(s->huff_up);
    if (s->huff_c2) (void)hipStreamDestroy(s->huff_c2);
    for (auto &e : s->huff_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    s->huff_up = s->huff_c2 = nullptr;
    s->plan = s->groups2 = nullptr;
    s->side = nullptr; s->fork = s->join = s->mid = s->aux = nullptr; s->device = -1;
}
struct SideHolder {
    SideStream s;
    SideHolder() { std::lock_guard<std::mutex> l(g_side_mu); g_sides.push_back(&s); }
    ~SideHolder()
    {
        std::lock_guard<std::mutex> l(g_side_mu);
        g_sides.erase(std::remove(g_sides.begin(), g_sides.end(), &s), g_sides.end());
        side_release(&s);
    }
};
SideStream *side_stream_for_this_thread()
{
    static thread_local SideHolder h;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> l(g_side_mu);
    if (h.s.side && h.s.device != dev) side_release(&h.s);
    if (!h.s.side) {
        /* the highest priority the device has: what runs here is the SHORT chain next to a large kernel of the caller's stream (the HEVC planner's
         * ticket kernels next to the per-pixel programs: a few workgroups each, which otherwise queue behind thousands), or, in the VP8
         * side-by-side call, a kernel whose share of the residency is its own (FFHIP_SIDE_PRIORITY=0: the default priority) */
        int least = 0, greatest = 0;
        const bool high = !FFHIP_ENV("FFHIP_SIDE_PRIORITY") || atoi(FFHIP_ENV("FFHIP_SIDE_PRIORITY")) != 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        if (hipStreamCreateWithPriority(&h.s.side, hipStreamNonBlocking, high ? greatest : 0) != hipSuccess || hipEventCreateWithFlags(&h.s.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h.s.join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h.s.mid, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h.s.aux, hipEventDisableTiming) != hipSuccess) {
            side_release(&h.s);
            return nullptr;
        }
        h.s.device = dev;
    }
    return &h.s;
}
} // namespace
/* the same side stream for other stages of the library that have two independent chains in one call (ffhip_hevc_intra_recon: the
 * substitution table next to the planner's kernels); 0 on success */
extern "C" int ffhip_side_stream_get(FfhipSide *out)
{
    SideStream *ss = side_stream_for_this_thread();
    if (!ss) return FFHIP_EIO;
    out->stream = ss->side; out->fork = ss->fork; out->join = ss->join; out->mid = ss->mid; out->aux = ss->aux;
    return FFHIP_OK;
}
/* ... and the streams and events of a pipelined call (ffhip_hevc_intra_recon_tiles), next to the thread's side stream and released with it */
extern "C" int ffhip_pipe_streams_get(FfhipPipe *out)
{
    SideStream *ss = side_stream_for_this_thread();
    if (!ss) return FFHIP_EIO;
    std::lock_guard<std::mutex> l(g_side_mu);
    if (!ss->plan) {
        bool ok = hipStreamCreateWithFlags(&ss->plan, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&ss->groups2, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; ok && k < FFHIP_PIPE_EVENTS; k++) ok = hipEventCreateWithFlags(&ss->pev[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (ss->plan) (void)hipStreamDestroy(ss->plan);
            if (ss->groups2) (void)hipStreamDestroy(ss->groups2);
            for (auto &e : ss->pev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
            ss->plan = ss->groups2 = nullptr;
            return FFHIP_EIO;
        }
    }
    out->plan = ss->plan; out->groups2 = ss->groups2;
    for (int k = 0; k < FFHIP_PIPE_EVENTS; k++) out->ev[k] = ss->pev[k];
    return FFHIP_OK;
}
extern "C" int ffhip_huff_streams_get(FfhipHuffStreams *out)
{
    SideStream *ss = side_stream_for_this_thread(); /* (recreated, and these with it, when the thread's current device has changed) */
    if (!ss) return FFHIP_EIO;
    std::lock_guard<std::mutex> l(g_side_mu);
    if (!ss->huff_up) {
        hipStream_t up = nullptr, c2 = nullptr;
        hipEvent_t ev[FFHIP_HUFF_PARTS + 4] = {};
        bool ok = hipStreamCreateWithFlags(&up, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c2, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; ok && k < FFHIP_HUFF_PARTS + 2; k++) ok = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
        for (int k = FFHIP_HUFF_PARTS + 2; ok && k < FFHIP_HUFF_PARTS + 4; k++) ok = hipEventCreate(&ev[k]) == hipSuccess;
        if (!ok) { /* nothing half-made is kept: the next call tries again from nothing */
            (void)hipGetLastError();
            if (up) (void)hipStreamDestroy(up);
            if (c2) (void)hipStreamDestroy(c2);
            for (auto &e : ev) if (e) (void)hipEventDestroy(e);
            return FFHIP_EIO;
        }
        for (int k = 0; k < FFHIP_HUFF_PARTS + 4; k++) ss->huff_ev[k] = ev[k];
        ss->huff_c2 = c2;
        ss->huff_up = up; /* last */
    }
    out->up = ss->huff_up; out->c2 = ss->huff_c2;
    for (int k = 0; k < FFHIP_HUFF_PARTS; k++) out->part_ev[k] = ss->huff_ev[k];
    out->fork = ss->huff_ev[FFHIP_HUFF_PARTS]; out->join = ss->huff_ev[FFHIP_HUFF_PARTS + 1];
    out->time_ev[0] = ss->huff_ev[FFHIP_HUFF_PARTS + 2]; out->time_ev[1] = ss->huff_ev[FFHIP_HUFF_PARTS + 3];
    return FFHIP_OK;
}
extern "C" void ffhip_vp8_retry_release(void);
extern "C" void ffhip_vp8_release_side_streams(void) /* ffhip_shutdown: nothing of the library's is in flight */
{
    ffhip_vp8_retry_release();
    std::lock_guard<std::mutex> l(g_side_mu);
    for (SideStream *s : g_sides) side_release(s);
}

/* ---- the side-by-side call is repeated by ffhip_stream_sync when it could not finish ----
 * What the prediction reads of the planes' FORMER contents is one column: the reference's wrapped 16x16 H_PRED at x = 0 reads the sample left
 * of a row's first pixel, i.e. the plane's last column a line further up (predict.c:346-353).  The one-call form keeps a copy of that column
 * (one byte per picture line) and a record of its arguments per stream; its two kernels report a bounded wait that ran out in a pinned word
 * of the RECORD's own -- so no other stream's or kernel's abort can set the retry off, and a record whose call ran clean never fires --; when
 * that word is set, ffhip_stream_sync puts the column back and runs prediction, then filter (then the colour conversion, for
 * ffhip_vp8_decode_frames' row form), one after the other -- everything else the stages read is their inputs, which are intact.  The sync
 * then says FFHIP_RETRIED, not FFHIP_OK: the planes hold the bytes of an undisturbed call, but what the CALLER enqueued behind the call has
 * consumed the aborted run's (round 4 said FFHIP_OK, and converted the aborted planes to BGRA in ffhip_vp8_decode_frames' row form). */
#define SCRATCH_VP8_RETRY 8
__global__ __launch_bounds__(256) void k_vp8_last_column(uint8_t *y, long long plane_y, int ys, int rows, int n_images, uint8_t *keep, int restore)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * n_images) return;
    const long long img = i / rows, r = i - img * rows;
    uint8_t *p = y + img * plane_y + r * ys + (ys - 1);
    if (restore) *p = keep[i];
    else keep[i] = *p;
}
thread_local FfhipVp8Then g_ffhip_vp8_then = {0, nullptr, 0, 0};
namespace {
struct Vp8Retry { /* one per stream, kept (the mode copy's storage and the pinned word are reused call after call); `armed` says whether it describes a call */
    bool armed = false;
    int mbcols = 0, mbrows = 0, n_images = 0, filter_type = 0;
    std::vector<uint8_t> h_modes;
    const uint8_t *d_modes = nullptr, *d_filters = nullptr;
    const int16_t *d_residual = nullptr;
    int64_t residual_stride = 0, plane_y = 0, plane_uv = 0;
    const int32_t *d_resmap = nullptr;
    uint8_t *y = nullptr, *u = nullptr, *v = nullptr, *keep = nullptr;
    FfhipVp8Then then = {0, nullptr, 0, 0};
    int *err = nullptr; /* pinned, device-visible: the call's own abort word */
    unsigned long long seq = 0;
};
std::mutex g_retry_mu;
std::map<void *, Vp8Retry> g_retry; /* by stream: the last side-by-side call enqueued there */
std::map<void *, unsigned long long> g_vp8_seq; /* by stream: VP8 prediction / filter calls enqueued so far -- a record is only good while its call is the LAST of them */
} // namespace
extern "C" void ffhip_vp8_note_enqueue(void *stream)
{
    std::lock_guard<std::mutex> l(g_retry_mu);
    g_vp8_seq[stream]++;
}
extern "C" void ffhip_vp8_retry_forget(void *stream) /* a clean ffhip_stream_sync: whatever was enqueued there has run */
{
    std::lock_guard<std::mutex> l(g_retry_mu);
    auto it = g_retry.find(stream);
    if (it != g_retry.end()) it->second.armed = false;
}
extern "C" void ffhip_vp8_retry_release(void) /* ffhip_shutdown */
{
    std::lock_guard<std::mutex> l(g_retry_mu);
    for (auto &e : g_retry)
        if (e.second.err) (void)hipHostFree(e.second.err);
    g_retry.clear();
    g_vp8_seq.clear();
}
/* ffhip_stream_sync(stream), the stream having drained: 0 = no side-by-side call of this stream reported anything; FFHIP_RETRIED = one had
 * run into a bounded wait, was repeated stage by stage and is done; FFHIP_EIO = it had, and could not be repeated (no longer the stream's
 * last VP8 call) or the repeat failed */
extern "C" int ffhip_vp8_side_by_side_retry(void *stream)
{
    Vp8Retry r;
    {
        std::lock_guard<std::mutex> l(g_retry_mu);
        auto it = g_retry.find(stream);
        if (it == g_retry.end() || !it->second.armed || !it->second.err) return 0;
        const int code = *(volatile int *)it->second.err;
        if (!code) return 0;
        *(volatile int *)it->second.err = 0;
        it->second.armed = false;
        if (it->second.seq != g_vp8_seq[stream]) return FFHIP_EIO; /* other VP8 calls went onto the stream behind it: their order cannot be restored */
        r = it->second;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long lines = (long long)16 * r.mbrows * r.n_images;
    hipLaunchKernelGGL(k_vp8_last_column, dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, st, r.y, (long long)r.plane_y, 16 * r.mbcols, 16 * r.mbrows, r.n_images, r.keep, 1);
    if (hipGetLastError() != hipSuccess) return FFHIP_EIO;
    int rc = ffhip_vp8_predict_recon(r.mbcols, r.mbrows, r.n_images, r.h_modes.data(), r.d_modes, r.d_residual, r.residual_stride, r.d_resmap, r.y, r.u, r.v,
                                     r.plane_y, r.plane_uv, stream);
    if (rc == FFHIP_OK) rc = ffhip_vp8_loopfilter(r.mbcols, r.mbrows, r.n_images, r.filter_type, r.d_modes, r.d_filters, r.y, r.u, r.v, r.plane_y, r.plane_uv, stream);
    if (rc == FFHIP_OK && r.then.on)
        rc = ffhip_yuv420_to_bgra(r.then.bgra, r.then.pitch, r.y, r.u, r.v, 16 * r.mbcols, 8 * r.mbcols, r.mbrows, r.mbcols, r.n_images, r.plane_y, r.plane_uv,
                                  r.then.image_stride, stream);
    if (rc != FFHIP_OK) return FFHIP_EIO;
    if (hipStreamSynchronize(st) != hipSuccess) return FFHIP_EIO;
    return FFHIP_RETRIED; /* (a wait of the repeat that ran out is in the process-wide word: the caller looks there next) */
}

/* Prediction + reconstruction and the loop filter of a batch of key frames as ONE call: both row kernels are enqueued
 * side by side (the filter on a stream of the library's own, forked behind the prediction's counter reset and joined back
 * into `stream`), the filter's rows following the prediction's through its per-row counters.  Same arguments and the same
 * bytes as ffhip_vp8_predict_recon followed by ffhip_vp8_loopfilter; the two chains overlap instead of adding up. */
extern "C" int ffhip_vp8_predict_recon(int mbcols, int mbrows, int n_images, const uint8_t *h_modes, const uint8_t *d_modes,
                                       const int16_t *d_residual, int64_t residual_stride, const int32_t *d_resmap, uint8_t *d_y,
                                       uint8_t *d_u, uint8_t *d_v, int64_t plane_stride_y, int64_t plane_stride_uv, void *stream);
extern "C" int ffhip_vp8_predict_loopfilter(int mbcols, int mbrows, int n_images, const uint8_t *h_modes, const uint8_t *d_modes,
                                            const int16_t *d_residual, int64_t residual_stride, const int32_t *d_resmap,
                                            int filter_type, const uint8_t *d_filters, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                                            int64_t plane_stride_y, int64_t plane_stride_uv, void *stream)
{
    /* taken and cleared before anything can return: an early return must not leave the record armed for the thread's next direct call */
    const FfhipVp8Then then = g_ffhip_vp8_then;
    g_ffhip_vp8_then.on = 0;
    if (filter_type < 0 || filter_type > 2) return FFHIP_EINVAL;
    if (filter_type != 0 && !d_filters) return FFHIP_EINVAL;
    const char *off = FFHIP_ENV("FFHIP_VP8_FUSE"); /* =0: one after the other on `stream` (A/B knob) */
    const bool fuse = filter_type != 0 && n_images > 0 && !(off && off[0] == '0') && ffhip_have_device();
    hipStream_t side = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    if (fuse) {
        SideStream *ss = side_stream_for_this_thread();
        if (!ss) return FFHIP_EIO;
        side = ss->side; fork_ev = ss->fork; join_ev = ss->join;
    }
    /* for the repeat: the last luma column as it is now, and who to call again (small batches only: the record holds a copy of the
     * host's mode bytes, which the row form checks on the host; a chip-filling batch leaves no room for the second kernel to be kept out) */
    uint8_t *keep = nullptr;
    const bool heal = fuse && h_modes && d_y && (long long)mbcols * mbrows * n_images <= (1LL << 17) && !FFHIP_ENV("FFHIP_VP8_NO_RETRY");
    int *err_word = nullptr;
    {
        std::lock_guard<std::mutex> l(g_retry_mu);
        Vp8Retry &r = g_retry[stream];
        if (r.err && *(volatile int *)r.err) { /* an earlier call's abort nobody collected (the caller never came back with ffhip_stream_sync): not lost */
            int *g = ffhip_async_err_word();
            if (g) *(volatile int *)g = *(volatile int *)r.err;
            *(volatile int *)r.err = 0;
        }
        r.armed = false;
        if (heal) {
            const long long lines = (long long)16 * mbrows * n_images;
            keep = (uint8_t *)ffhip_scratch(SCRATCH_VP8_RETRY, stream, (size_t)(lines + 3) / 4);
            if (!r.err && hipHostMalloc((void **)&r.err, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); r.err = nullptr; }
            if (keep && r.err) {
                *(volatile int *)r.err = 0;
                hipLaunchKernelGGL(k_vp8_last_column, dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_y, (long long)plane_stride_y, 16 * mbcols,
                                   16 * mbrows, n_images, keep, 0);
                if (hipGetLastError() == hipSuccess) {
                    r.mbcols = mbcols; r.mbrows = mbrows; r.n_images = n_images; r.filter_type = filter_type;
                    r.h_modes.assign(h_modes, h_modes + (size_t)mbcols * mbrows * n_images * 20); /* (the vector keeps its storage between calls) */
                    r.d_modes = d_modes; r.d_filters = d_filters; r.d_residual = d_residual; r.residual_stride = residual_stride; r.d_resmap = d_resmap;
                    r.plane_y = plane_stride_y; r.plane_uv = plane_stride_uv; r.y = d_y; r.u = d_u; r.v = d_v; r.keep = keep; r.then = then;
                    r.armed = true;
                    err_word = r.err;
                }
            }
        }
    }
    g_ffhip_vp8_fusion.err_word = err_word;
    g_ffhip_vp8_fusion.active = fuse ? 1 : 0;
    g_ffhip_vp8_fusion.pred_progress = nullptr;
    g_ffhip_vp8_fusion.side = side;
    g_ffhip_vp8_fusion.fork = fork_ev;
    int rc = ffhip_vp8_predict_recon(mbcols, mbrows, n_images, h_modes, d_modes, d_residual, residual_stride, d_resmap, d_y, d_u, d_v,
                                     plane_stride_y, plane_stride_uv, stream);
    const bool forked = fuse && g_ffhip_vp8_fusion.pred_progress != nullptr;
    if (rc == FFHIP_OK && filter_type != 0)
        rc = ffhip_vp8_loopfilter(mbcols, mbrows, n_images, filter_type, d_modes, d_filters, d_y, d_u, d_v, plane_stride_y, plane_stride_uv, stream);
    g_ffhip_vp8_fusion.active = 0;
    g_ffhip_vp8_fusion.pred_progress = nullptr;
    g_ffhip_vp8_fusion.err_word = nullptr;
    {
        std::lock_guard<std::mutex> l(g_retry_mu);
        auto it = g_retry.find(stream);
        if (it != g_retry.end() && it->second.armed) {
            if (rc == FFHIP_OK && forked) it->second.seq = g_vp8_seq[stream];
            else it->second.armed = false; /* nothing ran side by side: nothing to repeat */
        }
    }
    if (forked) { /* whatever happened to the filter's launch: `stream` continues behind the side stream */
        if (hipEventRecord(join_ev, side) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, join_ev, 0) != hipSuccess) return FFHIP_EIO;
    }
    return rc;
}
