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
 * TUs.  Default form: ONE launch, k_hevc_intra_groups (window groups, done flags per TU, see
 * there), scheduled by the device-side planner of ffhip_hevc_plan_gpu.hip or, for lists it
 * hands back, by plan_groups below.  Diagnostic form (FFHIP_HEVC_INTRA_MODE=levels): the host
 * gives each TU a wavefront level (1 + the highest level among the 4x4 blocks its available
 * neighbours lie in) and launches k_hevc_intra once per level, a wave per TU.
 * The TU body is the same in both: the 4n+1 neighbours live in LDS in
 * scan order (left column bottom-up, corner, top row left-to-right): in that order the
 * reference's substitution is "nearest available sample at or before me, else the first
 * available one", and its [1 2 1] smoothing is a 3-tap filter with untouched ends.
 */
#include "ffhip_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <map>
#include <memory>
#include <mutex>
#include <new>

#ifndef FFHIP_HEVC_INTRA_WAVES
#define FFHIP_HEVC_INTRA_WAVES 256 /* waves of the grouped form's one launch: enough for the widest wavefront of an 8K picture (~200 groups)
                                      and for a grid of 96 tiles; every further wave only holds a ticket far from its turn and polls
                                      (1024 waves: config-5 mix 6.5 ms, quadtree 11.7; 256: 6.0 / 11.4; tests/tools/bench_intra_c5.py) */
#endif
#ifndef FFHIP_HEVC_INTRA_WINDOW_LOG2
#define FFHIP_HEVC_INTRA_WINDOW_LOG2 5 /* luma window of the grouped form: 32x32 (1080p sweep in profiles/r1_stages.json) */
#endif

#define CTRL_HDR 352 /* words in front of the done flags: room for ten 128-byte lines wherever the block starts (ticket counters, abort) */
struct HevcIntraArgs {
    const ffhip_hevc_tu *tus;
    const uint32_t *work; /* TU indices of this level */
    const int16_t *residual;
    int16_t *plane[3];
    int stride[3];
    int bitdepth_y, bitdepth_c, count;
    /* grouped form only */
    const u32x4 *sched;       /* per schedule slot 3 x 16 B: the TU record (32 B), then {first wait entry,
                                 wait count | signal << 8 | tile_ok << 9, TU index, 0}               */
    const u32x4 *groups;      /* per group: {first slot, slot count, log2 window, 0}              */
    const uint32_t *wait_idx; /* TU indices a slot waits for (TUs of other groups)                */
    uint32_t *ctrl;           /* CTRL_HDR words, then one done flag per TU.  The header holds the next group ticket and the abort word, each ALONE in
                                 a 128-byte line (ctrl_ticket / ctrl_abort: their word indices): every wave takes its tickets from the one word with a
                                 device-scope atomic, and anything else that lives in its line queues up with them -- 826 resident waves that did nothing
                                 but read that line once in 30 us made a 135-tile grid take 1.58 ms instead of 1.08; the abort word, read by every
                                 waiting wave between two polls, sat next to it until late in round 4 */
    uint32_t ctrl_ticket, ctrl_abort;
    uint32_t ticket_shards;
    int *async_err;           /* pinned host word (ffhip_async_err_word)                          */
    int n_groups;
    int debug_withhold;       /* test hook (FFHIP_DEBUG_WITHHOLD_TU): this TU's done flag is never published; -1 = off */
    /* device-planned launches: the planner's verdict is read by the kernel, not by the host */
    const uint32_t *plan_result; /* {refused, number of groups, wait entries, -, widest wavefront}; NULL: n_groups above is the truth */
    uint32_t tp_width;           /* device-built plans: from this wavefront width on the throughput instance of the grouped kernel runs (0: never) */
    uint32_t poll_reps;          /* sleeps of ~0.5 us between two polls of a wave far from its turn (FFHIP_HEVC_INTRA_POLL_REPS) */
    uint32_t width_pct;          /* device-built plans: waves kept = widest wavefront x this / 128 (+ 16, at least 256): FFHIP_HEVC_INTRA_WIDTH_PCT */
    uint32_t wait_cap;           /* wait entries the planner had room for                                        */
    long long n_tus;             /* for the serial path a refused plan takes                                     */
    /* substitution table (k_hevc_intra_jtable): per TU and scan position the scan position its sample comes from */
    const uint8_t *jt;
    int jt_bw[3];                /* 4x4 blocks per row of each plane */
    uint32_t jt_boff[3];         /* first block of each plane        */
    /* per-pixel programs of the small TUs (k_hevc_intra_program): four LDS cell indices / weights per sample */
    const uint2 *desc;
    int desc_w[3];               /* samples per row of each plane in desc */
    uint32_t desc_off[3];        /* first entry of each plane             */
#ifdef FFHIP_INTRA_TRACE
    unsigned long long *trace;   /* diagnostics build only (make trace): 12 words per TU, then one per ticket */
#endif
};
#define JT_STRIDE 20 /* table bytes per 4x4 block: a TU of size n owns n/4 consecutive blocks of its first block row, 5n >= 4n + 1 bytes */
#ifdef FFHIP_INTRA_TRACE
static unsigned long long *g_intra_trace = nullptr;
extern "C" void ffhip_debug_intra_trace(void *d_buf) { g_intra_trace = (unsigned long long *)d_buf; }
#define TRACE_NOW() ((unsigned long long)wall_clock64())
__shared__ unsigned long long g_stamp[8];
#define STAMP(k) do { if (lane == 0) g_stamp[k] = TRACE_NOW(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int row_sum16(int v) /* every lane: the sum over its row of 16 lanes */
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);  /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);  /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false); /* row_half_mirror    */
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false); /* row_mirror         */
    return v;
}
/* v < 0 ? alt : v, as compare + select whatever the compiler thinks of the cost of computing alt */
__device__ __forceinline__ int neg_select(int v, int alt)
{
    int r;
    asm("v_cmp_gt_i32 vcc, 0, %1\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(v), "v"(alt) : "vcc");
    return r;
}
__device__ __forceinline__ int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

/* intraPredAngle and invAngle of 8.4.4.2.6 (the tables of format/predict.c) as ALU work on the wave-uniform mode:
 * a table in memory put a dependent global load (the angle) in front of every angular TU's reference gather and a
 * second one (the inverse angle) in front of its extension.  |angle| depends on the distance d of the mode from the
 * nearest of the pure directions 10 (horizontal) and 26 (vertical): {0, 2, 5, 9, 13, 17, 21, 26, 32}, positive on the
 * outer sides (modes 2-9 and 27-34); invAngle = -round(8192 / |angle|) for d = 1..8. */
__device__ __forceinline__ int intra_angle(int mode)
{
    const int d = mode < 18 ? (mode < 10 ? 10 - mode : mode - 10) : (mode < 26 ? 26 - mode : mode - 26);
    const unsigned long long mag = 0ull | (2ull << 6) | (5ull << 12) | (9ull << 18) | (13ull << 24) | (17ull << 30) | (21ull << 36) |
                                   (26ull << 42) | (32ull << 48);
    const int m = (int)((mag >> (6 * d)) & 63);
    return (mode < 10 || mode > 26) ? m : -m;
}
__device__ __forceinline__ int intra_inv_angle(int mode) /* modes 11..25 */
{
    const int d = mode < 18 ? mode - 10 : 26 - mode; /* 1..8 */
    const unsigned long long lo = 4096ull | (1638ull << 13) | (910ull << 26) | (630ull << 39);
    const unsigned long long hi = 482ull | (390ull << 13) | (315ull << 26) | (256ull << 39);
    const unsigned long long w = d <= 4 ? lo >> (13 * (d - 1)) : hi >> (13 * (d - 5));
    return -(int)(w & 8191);
}

/* `for (i = lane; i < LIMIT; i += 64)` with a compile-time LIMIT as a fixed number of predicated passes: the
 * loop form costs a compare, an exec-mask update and a branch per pass on the scalar unit, which a lone wave
 * (one instruction per ~9 cycles) feels; 32x32 blocks keep the loop where the passes are many */
#define LANE_PASSES(LIMIT) (((LIMIT) + 63) / 64)

#define NB_MAX 132 /* 4*32 + 1, padded */

/* The grouped form's LDS copy of its window, WITH a halo: cell (ty, tx) holds picture sample (wy0 - 1 + ty, wx0 - 1 + tx).
 * Row 0 -- the row above the window, with up to 64 more columns to the right for above-right neighbours -- has 130 cells of its
 * own; rows 1 .. 64 (the window, its left neighbour column in front) are 66 cells each: a neighbour to the right of the window in
 * one of THOSE rows belongs to a window that comes later in every coding-tree order, i.e. is never available (a list that claims
 * otherwise gets no cell from the program kernel and takes the generic body, which reads it from the plane).  The left column
 * continues below the window (below-left neighbours) in TILE_LEFT_EXT; three constant cells follow.  The window itself is written
 * by the wave as it reconstructs; halo cells are filled from memory by the TUs that need them (see intra_program).  One layout for
 * every window size.  (Round 2 kept 130 cells in every row: 17 KB of the kernel's 27 KB of LDS, which held the launch at five
 * waves per CU; with 9 KB it is the registers that bound it, at eight.) */
#define TILE_STRIDE 66 /* shorts per window row: 33 dwords, so a column walk hits 32 different banks */
#define TILE_ROW0 0 /* cells (0, tx), tx = 0 .. 129 */
#define TILE_BODY 132 /* cell (1, 0) */
#define TILE_ORIGIN (TILE_BODY + 1) /* cell of the window's first sample */
#define TILE_CELL(ty, tx) ((ty) == 0 ? TILE_ROW0 + (tx) : ((ty) <= 64 ? TILE_BODY + ((ty) - 1) * TILE_STRIDE + (tx) : TILE_LEFT_EXT + (ty) - 65)) /* any cell the layout has */
#define TILE_HAS_CELL(ty, tx) ((ty) == 0 ? (tx) <= 129 : ((ty) <= 64 ? (tx) <= TILE_STRIDE - 1 : ((ty) <= 128 && (tx) == 0)))
#define TILE_LEFT_EXT (TILE_BODY + 64 * TILE_STRIDE) /* cells (65 + k, 0), k = 0 .. 63 */
#define TILE_CONST_Y (TILE_LEFT_EXT + 64) /* 1 << (bitdepth_y - 1): what a TU without any neighbour predicts from */
#define TILE_CONST_C (TILE_LEFT_EXT + 65)
#define TILE_ZERO (TILE_LEFT_EXT + 66)
#define TILE_F (TILE_LEFT_EXT + 68) /* 33 cells: the smoothed neighbours of an 8x8 program (8.4.4.2.3), scan order */
#define TILE_CELLS (TILE_F + 34)

/* The residual block of a TU, fetched one TU ahead in the grouped form: samples 8 lane + 512 j .. +7 in v[j], as
 * 16-byte loads into registers of their own.  (The first form took one short per lane and pass into a short[16]:
 * the compiler packs two shorts to a VGPR, so every load was followed by s_waitcnt vmcnt(0) and a v_perm -- up to
 * sixteen exposed memory latencies per TU instead of a prefetch.)  Blocks that are not 16-byte aligned in
 * d_residual (odd offsets; no decoder lays residuals out like that) are read when they are used. */
struct ResPrefetch {
    u32x4 v[2];
    bool wide; /* wave-uniform: v[] holds the block */
};
__device__ __forceinline__ void fetch_residual(const HevcIntraArgs &a, const ffhip_hevc_tu &t, const int lane, ResPrefetch &rp)
{
    rp.wide = false;
    if (!(t.flags & 2)) return;
    const int nn = 1 << (2 * t.log2_size);
    const int16_t *src = a.residual + t.res_offset;
    if (((uintptr_t)src & 15) != 0) return;
    rp.wide = true;
#pragma unroll
    for (int j = 0; j < 2; j++)
        if (8 * lane + 512 * j < nn) rp.v[j] = *(const u32x4 *)(src + 8 * lane + 512 * j);
}

/* rdpcm accumulation of 8.6.5 on the wave's residual block in LDS.  Out of line on purpose: inlined four times (once
 * per TU size) its unrolled chunk arrays took the kernel to 314 VGPRs and a thousand register moves, and the scalar
 * registers that cost spilled in the common path; the path itself is rare (range extensions only). */
template <int LG>
__device__ __attribute__((noinline)) void rdpcm_accumulate(short *R, const int lane, const int mode)
{
    constexpr int n = 1 << LG;
    if (mode / 26 == 0) {
        /* running sum over the flattened block from index n (hevc.c:3963-3968):
         * R[i] = sum of R[n-1 .. i] mod 2^16 -- a prefix sum: per-lane chunks, then a wave scan of the chunk totals */
        constexpr int nn = n * n, chunk = nn >= 64 ? nn >> 6 : 1, active = nn / chunk;
        int loc[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            loc[j] = 0;
            if (j < chunk && lane < active) {
                const int idx = lane * chunk + j;
                sum += idx >= n - 1 ? (int)R[idx] : 0;
                loc[j] = sum;
            }
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const int excl = incl - sum;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < chunk && lane < active) {
                const int idx = lane * chunk + j;
                if (idx >= n - 1) R[idx] = (short)(loc[j] + excl);
            }
    } else if (lane < n) {
        for (int y = 1; y < n; y++) R[lane + n * y] = (short)(R[lane + n * y] + R[lane + n * (y - 1)]);
    }
}

/* One TU by one wave: steps 5-10 of decode_intra_block.  s, s2: NB_MAX ints each; refbase: 140 ints;
 * R: 32*32 shorts -- all private to the wave (LDS).
 * GROUPED: picture samples move with agent-scope accesses, and neighbours inside the group's window
 * (origin wx0, wy0, size 1 << wl) come from / go to the wave's LDS copy `tile` when tile_ok. */
template <bool GROUPED, int LG> /* LG = log2 of the TU size: every loop below has a compile-time trip count */
__device__ __forceinline__ void intra_tu(const HevcIntraArgs &a, const ffhip_hevc_tu &t, const int lane, int *s, int *s2,
                                         int *refbase, short *R, const ResPrefetch &rp, short *tile, const int wl,
                                         const bool tile_ok)
{
    const int wx0 = GROUPED ? (t.x >> wl) << wl : 0, wy0 = GROUPED ? (t.y >> wl) << wl : 0, wsz = GROUPED ? 1 << wl : 0;
    constexpr int n = 1 << LG, lg = LG;
    const int cidx = t.cidx, mode = t.pred_mode, flags = t.flags;
    const int bd = cidx == 0 ? a.bitdepth_y : a.bitdepth_c;
    /* selects, not a.plane[cidx]: indexing the kernel-argument arrays with a run-time index is two scalar memory
     * loads (and their latency) in front of every TU's gather */
    int16_t *plane = cidx == 0 ? a.plane[0] : (cidx == 1 ? a.plane[1] : a.plane[2]);
    const int stride = cidx == 0 ? a.stride[0] : (cidx == 1 ? a.stride[1] : a.stride[2]);
    const int x0 = t.x, y0 = t.y;
    constexpr int cnt = 4 * n + 1;
    int *ref = refbase + 34;

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

    /* ---- 1. gather + 2. substitute, in one pass: in scan order the substitution of 8.4.4.2.2 is
     * "the nearest available sample at or before me, else the first available one", so each
     * lane works out WHICH sample it wants from the masks alone and fetches that one ---- */
#pragma unroll
    for (int pass = 0; pass < LANE_PASSES(cnt); pass++) {
        const int i = lane + 64 * pass;
        if (i >= cnt) break;
        int j = i;
        if (n_avail < cnt && n_avail > 0) {
            j = -1;
            if (n == 32 && i >= 128 && m2) j = 128;
            if (n >= 16 && j < 0 && i >= 64) {
                const unsigned long long mm = i >= 127 ? m1 : (m1 & ((2ull << (i - 64)) - 1));
                if (mm) j = 127 - __clzll(mm);
            }
            if (j < 0) {
                const unsigned long long mm = i >= 63 ? m0 : (m0 & ((2ull << i) - 1));
                if (mm) j = 63 - __clzll(mm);
            }
            if (j < 0) j = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : 128);
        }
        int px, py;
        if (j < 2 * n) { px = x0 - 1; py = y0 + (2 * n - 1 - j); }
        else if (j == 2 * n) { px = x0 - 1; py = y0 - 1; }
        else { px = x0 + (j - 2 * n - 1); py = y0 - 1; }
        int v = 1 << (bd - 1);
        if (n_avail > 0) {
            const int16_t *sp = plane + (long long)py * stride + px;
            if (GROUPED) {
                const unsigned tx = (unsigned)(px - wx0), ty = (unsigned)(py - wy0);
                if (tile_ok && tx < (unsigned)wsz && ty < (unsigned)wsz) v = (int)tile[TILE_ORIGIN + ty * TILE_STRIDE + tx];
                else v = ffhip_load_s16_sc1(ffhip_rsrc(plane, 0xffffffffu), (py * stride + px) * 2);
            } else {
                v = (int)*sp;
            }
        }
        s[i] = v;
    }
    wave_sync();
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
#pragma unroll
            for (int pass = 0; pass < LANE_PASSES(cnt); pass++) {
                const int i = lane + 64 * pass;
                if (i >= cnt) break;
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
    const bool has_res = (flags & 2) != 0;
    if (has_res) {
        if (rp.wide) {
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (512 * j < n * n && 8 * lane + 512 * j < n * n) *(u32x4 *)(R + 8 * lane + 512 * j) = rp.v[j];
        } else {
            const int16_t *src = a.residual + t.res_offset;
            for (int i = lane; i < n * n; i += 64) R[i] = src[i];
        }
        wave_sync();
        if (flags & 0x40) {
            rdpcm_accumulate<LG>(R, lane, mode); /* rare (range extensions): kept out of line, see there */
            wave_sync();
        }
        if (flags & 0x80) { /* 8.6.6 with rY aliased to r, as at hevc.c:4753-4755; products wrap like -fwrapv */
            const int bdc = a.bitdepth_c, bdy = a.bitdepth_y;
#pragma unroll 4
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
        for (int i = 0; i < n; i++) sum += (unsigned)U16(LEFT(i)) + (unsigned)U16(TOP(i)); /* independent LDS reads: they pipeline (a wave
                                                                                              butterfly of dependent bpermutes measured slower) */
        dc = (int)((sum + (1u << lg)) >> (lg + 1));
    } else if (mode >= 2) {
        angle = intra_angle(mode);
        /* ref[] of 8.4.4.2.6: main = top for modes >= 18, left otherwise; both start at the corner */
#pragma unroll
        for (int pass = 0; pass < LANE_PASSES(2 * n + 1); pass++) {
            const int xx = lane + 64 * pass;
            if (xx > 2 * n) break;
            if (xx == 0) ref[0] = U16(TOP(-1));
            else if (xx <= n || angle >= 0) ref[xx] = mode >= 18 ? U16(TOP(xx - 1)) : U16(LEFT(xx - 1));
        }
        if (angle < 0 && ((n * angle) >> 5) < -1) {
            const int lo = (angle * n) >> 5, inv = intra_inv_angle(mode);
            const int xx = -1 - lane; /* lo >= -n >= -32: one pass */
            if (xx >= lo) {
                const int k = (xx * inv + 128) >> 8;
                ref[xx] = k == 0 ? U16(TOP(-1)) : (mode >= 18 ? U16(LEFT(k - 1)) : U16(TOP(k - 1)));
            }
        }
        wave_sync();
    }
    const bool edge_ok = cidx == 0 && n < 32;
    constexpr int pred_unroll = n * n <= 256 ? LANE_PASSES(n * n) : 4; /* 32x32: four of the sixteen passes in flight, so that their LDS reads overlap */
#pragma unroll pred_unroll
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
        int16_t *dp = plane + (long long)(y0 + y) * stride + x0 + x;
        const short rec = (short)clip3i(0, (1 << bd) - 1, pr + rs);
        if (GROUPED) {
            __hip_atomic_store(dp, rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned tx = (unsigned)(x0 + x - wx0), ty = (unsigned)(y0 + y - wy0);
            if (tx < (unsigned)wsz && ty < (unsigned)wsz) tile[TILE_ORIGIN + ty * TILE_STRIDE + tx] = rec;
        } else {
            *dp = rec;
        }
    }
#undef LEFT
#undef TOP
#undef U16
}

/* ---- the grouped form's TU body -------------------------------------------------------------------------------
 * Same arithmetic as intra_tu above, with everything that does not depend on SAMPLES taken off the wave's
 * instruction stream -- a lone wave pays ~6-9 cycles per instruction, and the critical path of a picture is ~10 000
 * TUs walked one after the other (tests/tools/diag_intra_trace.py):
 *   - which sample a scan position takes (availability masks, substitution) comes from a table a parallel kernel
 *     wrote beforehand (k_hevc_intra_jtable), fetched one TU ahead like the residual;
 *   - the angular predictor reads the scan-order array directly: ref[k] of 8.4.4.2.6 is s[2n + k] (k >= 0) or
 *     s[2n - ((k * invAngle + 128) >> 8)] (k < 0) for the vertical modes and the mirror image for the horizontal
 *     ones, so no ref[] array is built (that was an LDS write, a barrier and a dependent read per TU). */
struct IntraSlot {
    unsigned x, y, lg, cidx, mode, flags, res_offset;
    int res_scale;
    unsigned wait_begin, wait_count, signal, tile_ok, tu_index, jt_off;
};
struct JPrefetch {
    unsigned j[3];
};
/* what is the same for every TU of a group (one window of one plane) */
struct GroupCtx {
    __amdgpu_buffer_rsrc_t plane_rs;
    int cidx, stride, maxv, wl, wx0, wy0;
    int plane_lane[2], desc_lane[2]; /* per lane, 4x4 / 8x8: byte offset of the lane's pixel from the TU's first one */
};
/* Kernel arguments the grouped kernel touches for every TU, as values the compiler cannot re-derive: with the scalar
 * registers it has left it reloaded them from the kernel-argument segment where they were used -- a scalar memory
 * load and an s_waitcnt lgkmcnt(0) (which also drains the LDS queue) per use, ~0.8 us per generic TU in all. */
struct HotArgs { /* global-memory pointers by type: after the trip through a register the compiler would assume flat addressing */
    const __attribute__((address_space(1))) int16_t *residual;
    const __attribute__((address_space(1))) uint8_t *jt;
    const __attribute__((address_space(1))) uint32_t *wait_idx;
    int bitdepth_y, bitdepth_c;
};
__device__ __forceinline__ void fetch_jtable(const HotArgs &a, const IntraSlot &t, const int lane, JPrefetch &jp)
{
    const __attribute__((address_space(1))) uint8_t *p = a.jt + t.jt_off + lane; /* the table is padded: reading past a small TU's entries is harmless */
    jp.j[0] = p[0];
    if (t.lg >= 4) {
        jp.j[1] = p[64];
        jp.j[2] = p[128];
    }
}
__device__ __forceinline__ void fetch_residual_g(const HotArgs &a, const IntraSlot &t, const int lane, ResPrefetch &rp)
{
    rp.wide = false;
    if (!(t.flags & 2)) return;
    const int nn = 1 << (2 * t.lg);
    const __attribute__((address_space(1))) int16_t *src = a.residual + t.res_offset;
    if (((uintptr_t)src & 15) != 0) return;
    rp.wide = true;
#pragma unroll
    for (int j = 0; j < 2; j++)
        if (8 * lane + 512 * j < nn) rp.v[j] = *(const __attribute__((address_space(1))) u32x4 *)(src + 8 * lane + 512 * j);
}

template <int LG, class MID>
__device__ __forceinline__ void intra_tu_g(const HotArgs &a, const GroupCtx &g, const IntraSlot &t, const int lane, int *__restrict__ s, int *__restrict__ s2, short *__restrict__ R,
                                           const ResPrefetch &rp, const JPrefetch &jp, short *__restrict__ tile, const short *__restrict__ zero_block, MID &&mid)
{
    constexpr int n = 1 << LG, lg = LG, cnt = 4 * n + 1;
    const int x0 = (int)t.x, y0 = (int)t.y;
    const int wx0 = g.wx0, wy0 = g.wy0, wsz = 1 << g.wl;
    const int cidx = g.cidx, mode = (int)t.mode, flags = (int)t.flags;
    const bool tile_ok = t.tile_ok != 0;
    const int bd = cidx == 0 ? a.bitdepth_y : a.bitdepth_c;
    const int stride = g.stride;
    const __amdgpu_buffer_rsrc_t prs = g.plane_rs;
    /* ---- 1 + 2. gather with the substitution folded in; the samples stay in registers (scan position 64 * pass + lane)
     * until the smoothing has been applied: through LDS (write, read three, write again, two barriers and a swap of the
     * two scratch arrays that made every later address dynamic) the smoothing cost 0.55 - 0.65 us per TU ---- */
    constexpr int P = LANE_PASSES(cnt);
    int gv[P];
#pragma unroll
    for (int pass = 0; pass < P; pass++) {
        const int i = lane + 64 * pass;
        int v = 1 << (bd - 1);
        if (i < cnt) {
            const int j = (int)jp.j[pass];
            int px = x0 - 1, py = y0 - 1; /* the corner, j == 2n */
            if (j < 2 * n) py = y0 + (2 * n - 1 - j);
            else if (j > 2 * n) px = x0 + (j - 2 * n - 1);
            if (j != 255) { /* 255: nothing available around this TU (wave-uniform) */
                const unsigned tx = (unsigned)(px - wx0), ty = (unsigned)(py - wy0);
                if (tile_ok && tx < (unsigned)wsz && ty < (unsigned)wsz) v = (int)tile[TILE_ORIGIN + ty * TILE_STRIDE + tx];
                else v = ffhip_load_s16_sc1(prs, (py * stride + px) * 2);
            }
        }
        gv[pass] = v;
    }
    STAMP(0);
    mid(); /* the caller's fetches for the NEXT TU: behind this TU's own neighbour loads, with the rest of the TU to arrive in */
#define LEFT(y) s[2 * n - 1 - (y)]
#define TOP(x) s[2 * n + 1 + (x)] /* TOP(-1) is the corner */

    /* ---- 3. neighbour smoothing (8.4.4.2.3), across lanes: position i - 1 / i + 1 is the lane below / above (DPP
     * wave_shr / wave_shl), across a pass boundary lane 63 of the pass before / lane 0 of the pass behind ---- */
    bool smooth = false;
    if ((flags & 4) && mode != 1 && n != 4) {
        const int d26 = iabs(mode - 26), d10 = iabs(mode - 10);
        const int thr = n == 8 ? 7 : (n == 16 ? 1 : 0);
        smooth = (d26 < d10 ? d26 : d10) > thr;
    }
    if (smooth) {
        bool bi = false;
        int corner = 0, l63 = 0, t63 = 0;
        if (n == 32 && (flags & 8) && cidx == 0) { /* TOP(-1) = position 64, TOP(63) = 128, TOP(31) = 96, LEFT(63) = 0, LEFT(31) = 32 */
            corner = __builtin_amdgcn_readlane(gv[P > 1 ? 1 : 0], 0); l63 = __builtin_amdgcn_readlane(gv[0], 0); t63 = __builtin_amdgcn_readlane(gv[P - 1], 0);
            const int l31 = __builtin_amdgcn_readlane(gv[0], 32), t31 = __builtin_amdgcn_readlane(gv[P > 1 ? 1 : 0], 32);
            bi = iabs(corner + t63 - 2 * t31) < (1 << (a.bitdepth_y - 5)) && iabs(corner + l63 - 2 * l31) < (1 << (a.bitdepth_y - 5));
        }
#pragma unroll
        for (int pass = 0; pass < P; pass++) {
            const int i = lane + 64 * pass, g0 = gv[pass];
            int lo = __builtin_amdgcn_update_dpp(0, g0, 0x138, 0xf, 0xf, false); /* wave_shr:1 -- lane i - 1 */
            int hi = __builtin_amdgcn_update_dpp(0, g0, 0x130, 0xf, 0xf, false); /* wave_shl:1 -- lane i + 1 */
            if (pass > 0) lo = lane == 0 ? __builtin_amdgcn_readlane(gv[pass - 1], 63) : lo;
            if (pass + 1 < P) hi = lane == 63 ? __builtin_amdgcn_readlane(gv[pass + 1 < P ? pass + 1 : pass], 0) : hi;
            int f = (i == 0 || i == cnt - 1) ? g0 : (int)(short)((lo + 2 * g0 + hi + 2) >> 2);
            if (n == 32 && bi) {
                if (i < 2 * n) { const int y = 2 * n - 1 - i; f = y == 63 ? l63 : (corner * (63 - y) + (y + 1) * l63 + 32) >> 6; }
                else if (i == 2 * n) f = corner;
                else { const int x = i - 2 * n - 1; f = x == 63 ? t63 : (corner * (63 - x) + (x + 1) * t63 + 32) >> 6; }
                f = (int)(short)f;
            }
            if (i < cnt) s[i] = f;
        }
    } else {
#pragma unroll
        for (int pass = 0; pass < P; pass++) {
            const int i = lane + 64 * pass;
            if (i < cnt) s[i] = gv[pass];
        }
    }
    wave_sync();
    STAMP(1);

    /* ---- residual (with the optional rdpcm accumulation of 8.6.5) ---- */
    const bool has_res = (flags & 2) != 0;
    if (has_res) {
        if (rp.wide) {
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (512 * j < n * n && 8 * lane + 512 * j < n * n) *(u32x4 *)(R + 8 * lane + 512 * j) = rp.v[j];
        } else {
            const __attribute__((address_space(1))) int16_t *src = a.residual + t.res_offset;
            for (int i = lane; i < n * n; i += 64) R[i] = src[i];
        }
        wave_sync();
        if (flags & 0x40) {
            rdpcm_accumulate<LG>(R, lane, mode);
            wave_sync();
        }
        if (flags & 0x80) {
            const int bdc = a.bitdepth_c, bdy = a.bitdepth_y;
#pragma unroll 4
            for (int i = lane; i < n * n; i += 64) {
                const int up = (int)((unsigned)(int)R[i] << bdc) >> bdy;
                R[i] = (short)(R[i] + ((int)((unsigned)t.res_scale * (unsigned)up) >> 3));
            }
            wave_sync();
        }
    }

    STAMP(2);
    /* ---- 4 + 5. prediction and reconstruction, 64 samples a pass (the reference reads the neighbours as uint16_t and
     * stores the prediction as int16 before the add).  One loop per KIND of mode, chosen once: the first form decided
     * planar / DC / angular, the direction and the sign of the angle again for every pass, with 32-bit multiplies by
     * +-1 and 64-bit address arithmetic in between -- ~75 instructions and several taken branches a pass.  Here a pass
     * of the angular loop is ~30 instructions: what changes from pass to pass moves by additions (rows per pass are
     * fixed), 24-bit multiplies, tile and plane offsets by immediate / one add. ---- */
#define U16(v) ((int)((unsigned)(v) & 0xffffu))
    {
        constexpr int passes = n * n >= 64 ? n * n / 64 : 1, rows = n >= 8 ? 64 / n : 4; /* rows of the block per pass */
        constexpr int unroll = passes > 4 ? 4 : passes; /* 32x32: sixteen passes, four to a loop body (the kernel's code must stay inside the instruction cache) */
        const bool edge_ok = cidx == 0 && n < 32;
        const int maxv = (1 << bd) - 1;
        const int x = lane & (n - 1), yl = lane >> lg; /* the lane's sample in pass 0; later passes: y = yl + pass * rows */
        int goff = ((y0 + yl) * stride + x0 + x) * 2;  /* plane byte offset, stepped by gstep per pass */
        const int gstep = rows * stride * 2;
        /* the window copy is written without asking whether the sample lies in the window: a TU larger than its window
         * (it then starts at the window's origin) spills into cells right of and below the window that nobody reads --
         * halo cells are row 0 and column 0 only -- and the layout has room for a 32x32 block from any window origin */
        short *const cellp = tile + TILE_ORIGIN + (y0 - wy0 + yl) * TILE_STRIDE + (x0 - wx0 + x);
        const short *Rr = has_res ? R + lane : zero_block + lane; /* no residual: 64 zeros every pass reads again, no branch per pass */
        const int rstep = has_res ? 64 : 0;
#define EMIT(v_, pass_) do { \
            const int pr_ = (int)(short)((v_) & 0xffff); \
            const short rec_ = (short)clip3i(0, maxv, pr_ + (int)*Rr); \
            Rr += rstep; \
            __builtin_amdgcn_raw_buffer_store_b16(rec_, prs, goff, 0, FFHIP_AUX_SC1); \
            goff += gstep; \
            cellp[(pass_) * rows * TILE_STRIDE] = rec_; \
        } while (0)
        if (n >= 8 || lane < 16) {
            if (mode == 0) {
                const int tn = U16(TOP(n)), ln = U16(LEFT(n)), tx = U16(TOP(x));
                const int fixed = (x + 1) * tn + n; /* the part of the sum that does not change from pass to pass */
#pragma unroll unroll
                for (int pass = 0; pass < passes; pass++) {
                    const int y = yl + pass * rows;
                    const int v = ((n - 1 - x) * U16(LEFT(y)) + fixed + (n - 1 - y) * tx + (y + 1) * ln) >> (lg + 1);
                    EMIT(v, pass);
                }
            } else if (mode == 1) {
                /* the 2n-sample sum: one element per lane, added across the wave */
                const int e = lane < n ? U16(LEFT(lane)) : (lane < 2 * n ? U16(TOP(lane - n)) : 0);
                const int rsum = row_sum16(e); /* n = 4: lanes 8 .. 15 hold 0 */
                int sum = __builtin_amdgcn_readlane(rsum, 0);
                if (n >= 16) sum += __builtin_amdgcn_readlane(rsum, 16);
                if (n == 32) sum += __builtin_amdgcn_readlane(rsum, 32) + __builtin_amdgcn_readlane(rsum, 48);
                const int dc = (sum + n) >> (lg + 1);
                if (edge_ok && !(flags & 0x20)) {
                    const int tx = U16(TOP(x));
#pragma unroll unroll
                    for (int pass = 0; pass < passes; pass++) {
                        const int y = yl + pass * rows;
                        int v = dc;
                        if (x == 0) v = y == 0 ? (U16(LEFT(0)) + 2 * dc + tx + 2) >> 2 : (U16(LEFT(y)) + 3 * dc + 2) >> 2;
                        else if (pass == 0 && y == 0) v = (tx + 3 * dc + 2) >> 2;
                        EMIT(v, pass);
                    }
                } else {
#pragma unroll unroll
                    for (int pass = 0; pass < passes; pass++) EMIT(dc, pass);
                }
            } else {
                const int angle = intra_angle(mode);
                const bool vert = mode >= 18; /* the top row is the main reference: scan positions grow with the ref index */
                const int sg4 = vert ? 4 : -4; /* bytes per step of the ref index in the scan-order array */
                int al = vert ? yl : x, ac = vert ? x : yl; /* along / across the direction */
                const int dal = vert ? rows : 0, dac = vert ? 0 : rows;
                const char *const sc = (const char *)(s + 2 * n); /* the corner: ref[k] is at sc + sg4 * k for k >= 0 */
#define TAP(q_) U16(*(const int *)(sc + __mul24((q_), sg4)))
                const bool e26 = edge_ok && !(flags & 0x10) && mode == 26, e10 = edge_ok && !(flags & 0x10) && mode == 10;
                if (e26 || e10) { /* pure vertical / horizontal with the boundary filter (angle 0: one tap) */
                    const int corner = U16(TOP(-1)), first = e26 ? U16(TOP(0)) : U16(LEFT(0));
#pragma unroll unroll
                    for (int pass = 0; pass < passes; pass++) {
                        const int y = yl + pass * rows;
                        int v = e26 ? U16(TOP(x)) : U16(LEFT(y));
                        if (e26 && x == 0) v = clip3i(0, (1 << a.bitdepth_y) - 1, first + ((U16(LEFT(y)) - corner) >> 1));
                        if (e10 && pass == 0 && y == 0) v = clip3i(0, (1 << a.bitdepth_y) - 1, first + ((U16(TOP(x)) - corner) >> 1));
                        EMIT(v, pass);
                    }
                } else if (angle >= 0) {
#pragma unroll unroll
                    for (int pass = 0; pass < passes; pass++) {
                        const int prod = __mul24(al + 1, angle), fact = prod & 31, k0 = ac + (prod >> 5) + 1;
                        /* fact == 0: the second tap has weight 0 (it may lie one past the array: any finite value does) */
                        const int v = ((32 - fact) * TAP(k0) + fact * TAP(k0 + 1) + 16) >> 5;
                        EMIT(v, pass);
                        al += dal;
                        ac += dac;
                    }
                } else {
                    const int inv = intra_inv_angle(mode);
#pragma unroll unroll
                    for (int pass = 0; pass < passes; pass++) {
                        const int prod = __mul24(al + 1, angle), fact = prod & 31, k0 = ac + (prod >> 5) + 1;
                        /* 8.4.4.2.6: negative indices come from the other side through the inverse angle */
                        /* both forms computed, one v_cndmask: as a conditional the compiler wrapped the multiply in an exec-mask
                         * region, which ends the basic block -- and with it any overlap of this pass's LDS reads with the next one's */
                        const int q0 = neg_select(k0, -((__mul24(k0, inv) + 128) >> 8));
                        const int q1 = neg_select(k0 + 1, -((__mul24(k0 + 1, inv) + 128) >> 8));
                        const int v = ((32 - fact) * TAP(q0) + fact * TAP(q1) + 16) >> 5;
                        EMIT(v, pass);
                        al += dal;
                        ac += dac;
                    }
                }
#undef TAP
            }
        }
#undef EMIT
    }
    STAMP(3);
#undef LEFT
#undef TOP
#undef U16
}

/* ---- small TUs as per-pixel programs ---------------------------------------------------------------------------
 * A lone wave pays ~8 cycles per instruction, and ~9 000 of the ~11 000 TUs on the critical path of an 8K picture are
 * 4x4 and 8x8 ones (tests/tools/diag_intra_trace.py): what such a TU costs is the NUMBER of instructions between
 * "my neighbours are there" and "my samples are there".  Everything that does not depend on sample VALUES -- which
 * neighbour a pixel reads (mode, angle, substitution), where that neighbour lives in the wave's LDS tile, its weight,
 * and every address the TU needs -- is worked out by a parallel kernel in front (k_hevc_intra_program): per pixel four
 * 16-bit words, per TU four dwords in the slot.  The pixel words and the pixel's own residual are fetched one TU ahead.
 * What is left on the chain: up to four LDS reads, a dozen VALU operations, one LDS and one global store.  TUs that
 * qualify: 4x4 and 8x8, no neighbour smoothing (never at 4x4; at 8x8 only planar and modes 2 / 18 / 34 smooth), no
 * rdpcm / cross-component residual, every neighbour inside the window written by this group.  Neighbours OUTSIDE the
 * window are copied into the tile's halo first. */
#define PK_KIND(p) ((p) & 7u)
#define PROG_GENERIC 0u
#define PROG_ANGULAR 1u      /* two taps: cells a0, a1, weight a2 of the second                                       */
#define PROG_PLANAR 2u       /* cells LEFT(y), TOP(x), TOP(n), LEFT(n)                                                */
#define PROG_DC 3u           /* a2 = the lane's element of the 2n-sample sum, a3 = boundary filter of this pixel      */
#define PROG_ANGULAR_EDGE 4u /* modes 10 / 26 with the boundary filter: lanes with a3 set take a0 + ((a1 - a2) >> 1) */
#define PK_LG3 8u
#define PK_RES 16u
#define PK_OUTSIDE 32u
#define PK_SIGNAL FFHIP_PK_SIGNAL
#define PK_WAIT FFHIP_PK_WAIT
#define PK_FILTER 1024u /* 8x8 program with neighbour smoothing: the words' fourth field is the lane's scan position source */
#define PK_END 512u /* the sentinel behind a chunk's last slot */
#define PK_SLOW FFHIP_PK_SLOW /* anything but a plain program: generic TU, a wait, halo cells */
#define PROG_NO_RESIDUAL FFHIP_PROG_NO_RESIDUAL /* res_off of a TU without residual = size of the residual buffer resource: reads 0 */
struct ProgSlot {
    unsigned packed;    /* kind, flags above, bits 31:16 = LDS byte address of the TU's first sample in the tile */
    unsigned res_off;   /* byte offset of the residual block                                                     */
    unsigned desc_off;  /* byte offset of the TU's first pixel words                                             */
    unsigned plane_off; /* byte offset of the TU's first sample in its plane                                     */
};
struct ProgPrefetch {
    uint2 d;
    short res; /* kept as loaded: a conversion at the load would wait for it there */
};
#define LDS_U16(byte_addr) ((int)*(const unsigned short *)((const char *)tile + (byte_addr)))
/* neighbour smoothing of an 8x8 program: lane i < 33 holds scan position i (its cell comes with the pixel words), the
 * [1 2 1] filter runs across lanes, the ends stay, and the result goes to the TILE_F cells the taps point at */
__device__ __forceinline__ void program_filter_step(short *tile, const int lane, const unsigned src_cell)
{
    const int g = lane < 33 ? (int)*(const short *)((const char *)tile + src_cell) : 0;
    const int lo = __builtin_amdgcn_update_dpp(0, g, 0x138, 0xf, 0xf, false); /* wave_shr:1 -- lane i - 1 */
    const int hi = __builtin_amdgcn_update_dpp(0, g, 0x130, 0xf, 0xf, false); /* wave_shl:1 -- lane i + 1 */
    const int f = (lane == 0 || lane == 32) ? g : (lo + 2 * g + hi + 2) >> 2;
    if (lane < 33) tile[TILE_F + lane] = (short)f;
}
template <int LG>
__device__ __forceinline__ void intra_program(const HotArgs &a, const GroupCtx &g, const ProgSlot &t, const int lane,
                                              const ProgPrefetch &pp, short *tile, const int cell_lane)
{
    constexpr int n = 1 << LG;
    const bool filt = LG == 3 && (t.packed & PK_FILTER) != 0;
    if (filt) program_filter_step(tile, lane, pp.d.y >> 16);
    if (LG == 3 || lane < n * n) {
        unsigned a0 = pp.d.x & 0xffffu, a1 = pp.d.x >> 16, a2 = pp.d.y & 0xffffu, a3 = pp.d.y >> 16;
        const unsigned kind = PK_KIND(t.packed);
        if (filt && kind == PROG_PLANAR) { /* LEFT(y), TOP(x), TOP(n), LEFT(n) among the smoothed cells */
            const int x = lane & (n - 1), y = lane >> LG;
            a0 = 2u * (unsigned)(TILE_F + 2 * n - 1 - y); a1 = 2u * (unsigned)(TILE_F + 2 * n + 1 + x);
            a2 = 2u * (unsigned)(TILE_F + 3 * n + 1); a3 = 2u * (unsigned)(TILE_F + n - 1);
        }
        const int r0 = LDS_U16(a0), r1 = LDS_U16(a1);
        int v;
        if (kind == PROG_ANGULAR) {
            v = ((32 - (int)a2) * r0 + (int)a2 * r1 + 16) >> 5;
        } else if (kind == PROG_PLANAR) {
            const int x = lane & (n - 1), y = lane >> LG;
            const int r2 = LDS_U16(a2), r3 = LDS_U16(a3);
            v = ((n - 1 - x) * r0 + (x + 1) * r2 + (n - 1 - y) * r1 + (y + 1) * r3 + n) >> (LG + 1);
        } else if (kind == PROG_DC) {
            const int dc = (__builtin_amdgcn_readfirstlane(row_sum16(LDS_U16(a2))) + n) >> (LG + 1);
            v = dc;
            if (a3 == 1) v = (r0 + 2 * dc + r1 + 2) >> 2;
            else if (a3 == 2) v = (r1 + 3 * dc + 2) >> 2;
            else if (a3 == 3) v = (r0 + 3 * dc + 2) >> 2;
        } else {
            const int r2 = LDS_U16(a2);
            const int vn = ((32 - (int)a2) * r0 + (int)a2 * r1 + 16) >> 5;
            const int ve = clip3i(0, (1 << a.bitdepth_y) - 1, r0 + ((r1 - r2) >> 1));
            v = a3 ? ve : vn;
        }
        const int pr = (int)(short)(v & 0xffff);
        const short rec = (short)clip3i(0, g.maxv, pr + (int)pp.res);
        *(short *)((char *)tile + (t.packed >> 16) + cell_lane) = rec;
        __builtin_amdgcn_raw_buffer_store_b16(rec, g.plane_rs, g.plane_lane[LG - 2], __builtin_amdgcn_readfirstlane((int)t.plane_off), FFHIP_AUX_SC1); /* a scalar: left to itself the compiler loops over its 'distinct values' */
    }
}
/* halo: every scan position of a program TU whose (substituted) source lies outside the window comes from memory --
 * its owner's done flag has been seen.  LDS serves a wave in order: the program's reads need no barrier */
__device__ __forceinline__ void intra_program_halo(const GroupCtx &g, const int x0, const int y0, const int n, const int lane,
                                                   const unsigned j, short *tile)
{
    const int wsz = 1 << g.wl;
    if (lane < 4 * n + 1 && j != 255u) {
        int px = x0 - 1, py = y0 - 1;
        if ((int)j < 2 * n) py = y0 + (2 * n - 1 - (int)j);
        else if ((int)j > 2 * n) px = x0 + ((int)j - 2 * n - 1);
        const int tx = px - g.wx0 + 1, ty = py - g.wy0 + 1;
        if ((tx == 0 || ty == 0 || tx > wsz || ty > wsz) && TILE_HAS_CELL(ty, tx)) { /* a source without a cell is one no pixel of this program reads (k_hevc_intra_program) */
            const int v = ffhip_load_s16_sc1(g.plane_rs, (py * g.stride + px) * 2);
            tile[TILE_CELL(ty, tx)] = (short)v;
        }
    }
}

/* The programs: one wave per schedule slot, in front of the grouped kernel.  Reads the substitution table, decides
 * whether the TU qualifies, writes the four words of every pixel and the slot's program words (second quarter of the
 * slot: the availability masks that lived there are in the substitution table now). */
struct ProgArgs {
    u32x4 *sched;
    uint32_t n_slots;
    const uint8_t *jt;
    uint2 *desc;
    int desc_w[3];
    uint32_t desc_off[3];
    int wl[3], stride[3];
    const uint32_t *plan_result;
    uint32_t wait_cap;
    /* "early" form (device-planned lists, slot = TU index): the slot's first and third quarter are not written yet -- the ticket kernels and
     * k_plan_emit run NEXT TO this kernel -- and are put together from what k_plan_count left: the TU record, its flag byte, its wait count */
    const ffhip_hevc_tu *tus;
    const uint8_t *flags;
    const uint32_t *wcount;
    int jt_bw[3];
    uint32_t jt_boff[3];
    const uint32_t *refused; /* NULL, or the word k_hevc_check_tus sets for a list with a bad record (then the records' positions mean nothing) */
    const ffhip_hevc_tu *records; /* the TU list, always (a slot's third quarter names its TU): the availability masks are read from it */
};
/* The program of ONE slot, by one wave.  Everything that differs from lane to lane is a SELECT, never a branch: the kernel was bound by
 * the scalar unit -- 286 scalar instructions per slot, most of them the exec-mask bookkeeping of per-lane ifs around the table reads
 * (252 M of them for the 1.84 M slots of an eight-picture grid, at one per cycle and CU: 0.47 ms) -- so a table read that a lane does not
 * need is made anyway, at position 0, and its answer dropped; on a 4x4 slot lanes 16 .. 63 do what lanes 0 .. 15 do (same words to the
 * same addresses).  What IS the same for all lanes (the mode's kind, the smoothing) stays a scalar branch, and what is the same for the
 * whole SLOT -- decoding the record, the angle tables, the window, where the slot's words go -- is not worked out here at all: the lane
 * that owns the slot has done that for its slot in the vector unit, 64 slots at a time (SlotPre), and the wave picks it up by readlane.
 * Returns the slot's verdict: bit 0 = every source has a cell in the tile layout (else the generic body), bit 1 = a source lies outside
 * the window (halo). */
struct SlotPre {
    uint32_t p0;      /* x0 | y0 << 16 */
    uint32_t p1;      /* bit 0: 8x8; 1-2: plane; 3-8: mode; 9: smoothing; 10: DC boundary filter; 11: modes 10 / 26 with theirs; 12-14: log2 of the window; 15: scan position 32 is available */
    uint32_t p2;      /* intraPredAngle + 32 | -invAngle << 8 */
    uint32_t m_lo;    /* availability of scan positions 0 .. 31 (left column bottom-up, corner, top row), position 32 in bit 15 of p1 */
    uint32_t d_first; /* index of the TU's first pixel words */
};
__device__ __forceinline__ unsigned intra_program_slot(const ProgArgs &a, const int lane, const uint32_t P0, const uint32_t P1, const uint32_t P2,
                                                       const uint32_t M_LO, const uint32_t DF)
{
    const int x0 = (int)(P0 & 0xffff), y0 = (int)(P0 >> 16);
    const int lg = 2 + (int)(P1 & 1u), cidx = (int)((P1 >> 1) & 3u), mode = (int)((P1 >> 3) & 63u);
    const bool filt = (P1 >> 9) & 1u, edge_dc = (P1 >> 10) & 1u, edge_tu = (P1 >> 11) & 1u;
    const int n = 1 << lg, wl = (int)((P1 >> 12) & 7u), wsz = 1 << wl;
    const int wx0 = (x0 >> wl) << wl, wy0 = (y0 >> wl) << wl;
    const uint32_t m_hi = (P1 >> 15) & 1u;
    const bool none = (M_LO | m_hi) == 0;                       /* nothing around the TU is available (the whole slot: a scalar) */
    const int first = M_LO ? __builtin_ctz(M_LO) : 32;           /* the first available scan position */
    const unsigned cconst = 2u * (cidx == 0 ? TILE_CONST_Y : TILE_CONST_C);
    bool ok = true, outside = false;
    /* LDS byte address of the sample scan position pos takes; a lane that does not `use` the answer reads position 0 and leaves the slot's
     * verdicts (ok, outside) alone */
    /* ... the substitution of 8.4.4.2.2 worked out here, from the availability bits: "the nearest available position at or before mine, else
     * the first available one" -- seven vector instructions where a byte of the substitution table was a trip to memory in front of every
     * program (the kernel waited 68 % of its wave cycles for those bytes once the scalar work was gone) */
    /* What scan position p takes is worked out ONCE per slot, by lane p (33 positions at most): the LDS byte address of its source cell
     * with two flags on top (bit 16: the source lies outside the window, bit 17: the tile layout has no cell for it).  A tap then is a
     * cross-lane read of that word (ds_bpermute) instead of the whole evaluation again -- two to four of them per pixel before. */
    unsigned tab;
    {
        const int pos = lane;
        const uint32_t upto = pos >= 31 ? ~0u : (2u << pos) - 1u;
        const uint32_t mm = M_LO & upto;
        const int jn = mm ? 31 - __builtin_clz(mm) : first;
        const int j = (pos >= 32 && m_hi) ? 32 : jn;
        const int py = y0 - 1 + max(2 * n - j, 0), px = x0 - 1 + max(j - 2 * n, 0); /* j < 2n: the left column, bottom-up; j > 2n: the top row */
        const int tx = px - wx0 + 1, ty = py - wy0 + 1;
        const bool out = ((unsigned)(tx - 1) >= (unsigned)wsz) | ((unsigned)(ty - 1) >= (unsigned)wsz);
        /* TILE_HAS_CELL / TILE_CELL with every arm worked out first: as nested conditional expressions they became nested branches */
        const bool z = ty == 0, m = ty <= 64;
        const bool has0 = tx <= 129, has1 = tx <= TILE_STRIDE - 1, has2 = (ty <= 128) & (tx == 0);
        const int cell0 = TILE_ROW0 + tx, cell1 = TILE_BODY + (ty - 1) * TILE_STRIDE + tx, cell2 = TILE_LEFT_EXT + ty - 65;
        const bool has12 = m ? has1 : has2, has = z ? has0 : has12;
        const int cell12 = m ? cell1 : cell2, cellw = z ? cell0 : cell12;
        const unsigned cw = has ? 2u * (unsigned)cellw : 0u;
        tab = none ? cconst : (cw | (out ? 1u << 16 : 0u) | (has ? 0u : 1u << 17));
    }
    auto src_cell = [&](const int pos, const bool use) -> unsigned {
        const unsigned v = (unsigned)__shfl((int)tab, use ? pos : 0, 64);
        outside = outside | (use & ((v >> 16) & 1u));
        ok = ok & (!use | !((v >> 17) & 1u));
        return v & 0xffffu;
    };
    auto cell = [&](const int pos, const bool use) -> unsigned { return filt ? 2u * (unsigned)(TILE_F + pos) : src_cell(pos, use); }; /* what a tap reads */
#define POS_LEFT(yy) (2 * n - 1 - (yy))
#define POS_TOP(xx) (2 * n + 1 + (xx))
    const int le = lane & (n * n - 1); /* the pixel this lane works for */
    const int x = le & (n - 1), y = le >> lg;
    unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (mode == 0) {
        if (!filt) { w0 = cell(POS_LEFT(y), true); w1 = cell(POS_TOP(x), true); w2 = cell(POS_TOP(n), true); w3 = cell(POS_LEFT(n), true); }
    } else if (mode == 1) {
        w0 = cell(POS_LEFT(y), true); w1 = cell(POS_TOP(x), true);
        const bool sum_lane = le < 2 * n; /* lanes 0 .. 2n - 1 hold the 2n samples the sum runs over */
        const unsigned ws = cell(le < n ? POS_LEFT(le) : POS_TOP(le - n), sum_lane);
        w2 = sum_lane ? ws : 2u * TILE_ZERO;
        if (edge_dc) w3 = (x == 0 && y == 0) ? 1u : (y == 0 ? 2u : (x == 0 ? 3u : 0u));
    } else {
        const int angle = (int)(P2 & 0xffu) - 32, inv = -(int)(P2 >> 8), sg = mode >= 18 ? 1 : -1;
        const int al = mode >= 18 ? y : x, ac = mode >= 18 ? x : y;
        const int prod = (al + 1) * angle, idx = prod >> 5, fact = prod & 31;
        const int k0 = ac + idx + 1, k1 = k0 + 1;
        const int p0 = k0 >= 0 ? k0 : -((k0 * inv + 128) >> 8), p1 = k1 >= 0 ? k1 : -((k1 * inv + 128) >> 8);
        const bool e = edge_tu && (mode == 26 ? x == 0 : y == 0); /* this pixel takes the boundary filter of modes 10 / 26 */
        const int posA = e ? (mode == 26 ? POS_TOP(0) : POS_LEFT(0)) : 2 * n + sg * p0;
        const int posB = e ? (mode == 26 ? POS_LEFT(y) : POS_TOP(x)) : 2 * n + sg * p1;
        w0 = cell(posA, true);
        const unsigned wb = cell(posB, e | (fact != 0)); /* weight 0: any cell does (the true one may lie past the array) */
        w1 = (e | (fact != 0)) ? wb : w0;
        if (edge_tu) { /* (wave-uniform) */
            const unsigned wc = cell(2 * n, e);
            w2 = e ? wc : (unsigned)fact;
            w3 = e ? 1u : 0u;
        } else {
            w2 = (unsigned)fact;
        }
    }
#undef POS_LEFT
#undef POS_TOP
    if (filt) { /* lane i fetches scan position i for the smoothing step */
        const bool fl = lane < 4 * n + 1;
        const unsigned wf = src_cell(lane, fl);
        w3 = fl ? wf : w3;
    }
    const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0, any_out = __builtin_amdgcn_ballot_w64(outside) != 0;
    if (all_ok) {
        const int dw = cidx == 0 ? a.desc_w[0] : (cidx == 1 ? a.desc_w[1] : a.desc_w[2]);
        a.desc[DF + (uint32_t)y * (uint32_t)dw + (uint32_t)x] = make_uint2(w0 | (w1 << 16), w2 | (w3 << 16));
    }
    return (all_ok ? 1u : 0u) | (any_out ? 2u : 0u);
}

/* A wave takes 64 consecutive slots: every lane decides for ITS slot whether it can be a program at all -- the answer is no for
 * the TUs above 8x8, rdpcm / cross-component TUs and TUs whose in-window neighbours another group wrote, i.e. for most of a
 * config-5 list -- and prepares what its slot's program needs (SlotPre, and the slot's program words but for the two bits the program
 * itself decides); then the wave builds the programs of the slots that remain, one after the other, and every lane writes its slot's words.
 * (One wave per slot, as before round 3, was bound by the rate waves can be started at: 0.75 ms for the 1.84 M slots of an
 * eight-picture grid, most of which left after reading two dwords.) */
__global__ __launch_bounds__(256) void k_hevc_intra_program(ProgArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64u;
    if (base >= a.n_slots) return;
    if (a.plan_result && (a.plan_result[0] | (a.plan_result[2] > a.wait_cap ? 1u : 0u))) return; /* refused plan: the slots mean nothing */
    if (a.refused && *a.refused) return;
    const uint32_t k = base + (uint32_t)lane;
    bool prog = false;
    SlotPre sp = {0u, 0u, 0u, 0u, 0u};
    u32x4 q1 = {PK_SLOW, PROG_NO_RESIDUAL, 0u, 0u}; /* the generic body */
    u32x4 q0 = {0, 0, 0, 0}, q2 = {0, 0, 0, 0};
    if (k < a.n_slots) {
        if (a.tus) {
            q0 = ((const u32x4 *)(a.tus + k))[0];
            const uint32_t f = a.flags[k], cidx = (q0.y >> 8) & 0xff;
            const uint32_t jb = cidx == 0 ? a.jt_boff[0] : (cidx == 1 ? a.jt_boff[1] : a.jt_boff[2]);
            const uint32_t jw = (uint32_t)(cidx == 0 ? a.jt_bw[0] : (cidx == 1 ? a.jt_bw[1] : a.jt_bw[2]));
            q2.y = a.wcount[k] | ((f & 1u) << 8) | (((f >> 1) & 1u) << 9);
            q2.z = k;
            q2.w = (jb + (q0.x >> 18) * jw + ((q0.x & 0xffffu) >> 2)) * JT_STRIDE;
        } else {
            q0 = a.sched[(size_t)k * 3]; q2 = a.sched[(size_t)k * 3 + 2];
        }
    }
    {
        const int x0 = (int)(q0.x & 0xffff), y0 = (int)(q0.x >> 16);
        const int lg = (int)(q0.y & 0xff), cidx = (int)((q0.y >> 8) & 0xff), mode = (int)((q0.y >> 16) & 0xff), flags = (int)(q0.y >> 24);
        const int wl = cidx == 0 ? a.wl[0] : (cidx == 1 ? a.wl[1] : a.wl[2]);
        const int n = 1 << lg;
        prog = k < a.n_slots && lg <= 3 && lg >= 2 && n <= (1 << wl) && ((q2.y >> 9) & 1) && !(flags & 0xC0) && q0.z < 0x7fff0000u && cidx <= 2;
        bool filt = false; /* neighbour smoothing applies (8.4.4.2.3; at 8x8: planar and modes 2, 18, 34): the taps read the smoothed copy */
        {
            const int d26 = iabs(mode - 26), d10 = iabs(mode - 10);
            filt = (flags & 4) && mode != 1 && n != 4 && (d26 < d10 ? d26 : d10) > 7;
        }
        const bool edge_ok = cidx == 0; /* n < 32 here */
        const bool edge_dc = edge_ok && !(flags & 0x20), edge_tu = edge_ok && !(flags & 0x10) && (mode == 26 || mode == 10);
        const bool ang = mode >= 2 && mode <= 34;
        const int angle = ang ? intra_angle(mode) : 0, inv = (ang && angle < 0) ? intra_inv_angle(mode) : 0;
        const int wx0 = (x0 >> wl) << wl, wy0 = (y0 >> wl) << wl;
        const uint32_t doff = cidx == 0 ? a.desc_off[0] : (cidx == 1 ? a.desc_off[1] : a.desc_off[2]);
        const int dw = cidx == 0 ? a.desc_w[0] : (cidx == 1 ? a.desc_w[1] : a.desc_w[2]);
        const int stride = cidx == 0 ? a.stride[0] : (cidx == 1 ? a.stride[1] : a.stride[2]);
        sp.p0 = q0.x;
        sp.p1 = (lg == 3 ? 1u : 0u) | ((uint32_t)cidx << 1) | ((uint32_t)(mode & 63) << 3) | (filt ? 1u << 9 : 0u) | (edge_dc ? 1u << 10 : 0u) | (edge_tu ? 1u << 11 : 0u) |
                ((uint32_t)(wl & 7) << 12);
        sp.p2 = (uint32_t)(angle + 32) | ((uint32_t)(-inv) << 8);
        {   /* availability by scan position, as k_hevc_intra_jtable lays it out: left column bottom-up (bit i = left[2n - 1 - i]), corner, top row */
            u32x4 av = {0u, 0u, 0u, 0u};
            if (prog) av = ((const u32x4 *)(a.records + q2.z))[1];
            const unsigned long long avail_top = (unsigned long long)av.x | ((unsigned long long)av.y << 32), avail_left = (unsigned long long)av.z | ((unsigned long long)av.w << 32);
            const int n2 = prog ? 2 * n : 8;
            const unsigned long long m0 = (__brevll(avail_left) >> (64 - n2)) | ((unsigned long long)(flags & 1) << n2) | ((avail_top & ((1ull << n2) - 1)) << (n2 + 1));
            sp.m_lo = (uint32_t)m0;
            sp.p1 |= (uint32_t)((m0 >> 32) & 1ull) << 15;
        }
        sp.d_first = doff + (uint32_t)y0 * (uint32_t)dw + (uint32_t)x0;
        if (prog) {
            const unsigned kind = mode == 0 ? PROG_PLANAR : (mode == 1 ? PROG_DC : (edge_tu ? PROG_ANGULAR_EDGE : PROG_ANGULAR));
            q1.x = kind | (lg == 3 ? PK_LG3 : 0u) | (filt ? PK_FILTER : 0u) | ((flags & 2) ? PK_RES : 0u) | (((q2.y >> 8) & 1) ? PK_SIGNAL : 0u) |
                   ((q2.y & 0xff) ? (PK_WAIT | PK_SLOW) : 0u) | ((uint32_t)(2 * (TILE_ORIGIN + (y0 - wy0) * TILE_STRIDE + (x0 - wx0))) << 16);
            q1.y = (flags & 2) ? q0.z * 2u : PROG_NO_RESIDUAL;
            q1.z = sp.d_first * 8u;
            q1.w = (uint32_t)(y0 * stride + x0) * 2u;
        }
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(prog);
    unsigned verdict = 1u; /* of MY slot: bit 0 = stays a program, bit 1 = reads halo cells */
    while (todo) {
        const int b = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned r = intra_program_slot(a, lane, (uint32_t)__builtin_amdgcn_readlane((int)sp.p0, b), (uint32_t)__builtin_amdgcn_readlane((int)sp.p1, b),
                                              (uint32_t)__builtin_amdgcn_readlane((int)sp.p2, b), (uint32_t)__builtin_amdgcn_readlane((int)sp.m_lo, b),
                                              (uint32_t)__builtin_amdgcn_readlane((int)sp.d_first, b));
        verdict = lane == b ? r : verdict;
    }
    if (k < a.n_slots) {
        const u32x4 generic = {PK_SLOW, PROG_NO_RESIDUAL, 0u, 0u};
        if (prog && (verdict & 2u)) q1.x |= PK_OUTSIDE | PK_SLOW;
        a.sched[(size_t)k * 3 + 1] = (prog && (verdict & 1u)) ? q1 : generic;
    }
}

template <class MID>
__device__ __forceinline__ void intra_tu_g_any(const HotArgs &a, const GroupCtx &g, const IntraSlot &t, const int lane, int *s, int *s2, short *R,
                                               const ResPrefetch &rp, const JPrefetch &jp, short *tile, const short *zero_block, MID &&mid)
{
    switch (t.lg) {
    case 2: intra_tu_g<2>(a, g, t, lane, s, s2, R, rp, jp, tile, zero_block, mid); break;
    case 3: intra_tu_g<3>(a, g, t, lane, s, s2, R, rp, jp, tile, zero_block, mid); break;
    case 4: intra_tu_g<4>(a, g, t, lane, s, s2, R, rp, jp, tile, zero_block, mid); break;
    default: intra_tu_g<5>(a, g, t, lane, s, s2, R, rp, jp, tile, zero_block, mid); break;
    }
}

/* The substitution table: for every TU and every scan position i (left column bottom-up, corner, top row) the scan
 * position j whose sample position i takes -- "the nearest available one at or before me, else the first available
 * one" (8.4.4.2.2 in scan order) -- or 255 when nothing around the TU is available.  A pure function of the TU record:
 * one wave per TU, fully parallel, in front of the dependency-bound kernel. */
struct JTabArgs {
    const ffhip_hevc_tu *tus;
    uint32_t n;
    uint8_t *jt;
    int bw[3];
    uint32_t boff[3];
    const uint32_t *refused; /* NULL, or the word k_hevc_check_tus sets for a list with a bad record (then the records' positions mean nothing) */
};
/* A LANE per TU: the table of a TU is a running scan over its positions -- j(i) = i where position i is available, else j(i - 1), starting
 * from the first available position -- two instructions a position, four positions to a stored dword (a TU's stripe has 5n >= 4n + 4 bytes),
 * and the lanes of a wave walk 64 tables side by side.  (A wave per TU was bound by the rate waves start at; a wave walking 64 TUs one after
 * the other, every lane a position, by its instruction count -- ~120 per TU, most of them scalar: 0.23 ms for the 1.84 M TUs of an
 * eight-picture grid, whose tables are 60 us worth of bytes.) */
__global__ __launch_bounds__(256) void k_hevc_intra_jtable(JTabArgs a)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (a.refused && *a.refused) return;
    if (i >= a.n) return;
    const u32x4 *rec = (const u32x4 *)(a.tus + i);
    const u32x4 r0 = rec[0], r1 = rec[1]; /* x | y << 16, log2_size | cidx << 8 | mode << 16 | flags << 24, res_offset, res_scale; avail_top, avail_left */
    const unsigned long long avail_top = (unsigned long long)r1.x | ((unsigned long long)r1.y << 32), avail_left = (unsigned long long)r1.z | ((unsigned long long)r1.w << 32);
    const int tx = (int)(r0.x & 0xffff), ty = (int)(r0.x >> 16), lg = (int)(r0.y & 0xff), cidx = (int)((r0.y >> 8) & 0xff), flags = (int)(r0.y >> 24);
    if (lg < 2 || lg > 5 || cidx > 2) return; /* (a list nobody has validated yet: k_hevc_check_tus refuses it) */
    const int n = 1 << lg;
    /* availability by scan position as a 129-bit number: left column bottom-up (bit i = left[2n - 1 - i]), corner (bit 2n), top row */
    const unsigned long long rl = __brevll(avail_left) >> (64 - 2 * n);
    const unsigned long long c = (flags & 1) ? 1ull : 0ull;
    const unsigned long long tp = n == 32 ? avail_top : (avail_top & ((1ull << (2 * n)) - 1));
    unsigned long long w0, w1;
    unsigned w2 = 0;
    if (n == 32) { /* left 0..63, corner 64, top 65..128 */
        w0 = rl; w1 = c | (tp << 1); w2 = (unsigned)(tp >> 63);
    } else {
        w0 = rl | (c << (2 * n)) | (tp << (2 * n + 1));
        w1 = tp >> (63 - 2 * n); /* bits that spill past 64 (n = 16: 4n + 1 = 65) */
    }
    const int first = w0 ? __builtin_ctzll(w0) : (w1 ? 64 + __builtin_ctzll(w1) : (w2 ? 128 : 255)); /* 255: nothing around the TU is available */
    const int bw = cidx == 0 ? a.bw[0] : (cidx == 1 ? a.bw[1] : a.bw[2]);
    const uint32_t boff = cidx == 0 ? a.boff[0] : (cidx == 1 ? a.boff[1] : a.boff[2]);
    uint32_t *out = (uint32_t *)(a.jt + (size_t)(boff + (uint32_t)(ty >> 2) * (uint32_t)bw + (uint32_t)(tx >> 2)) * JT_STRIDE); /* JT_STRIDE is a multiple of 4 */
    uint32_t j = (uint32_t)first;
    for (int q = 0; q <= n; q++) { /* 4n + 1 positions: n + 1 dwords */
        const unsigned long long word = q < 16 ? w0 : (q < 32 ? w1 : (unsigned long long)w2);
        const uint32_t nib = (uint32_t)(word >> (4 * (q & 15))) & 15u, p = 4u * (uint32_t)q;
        const uint32_t j0 = (nib & 1u) ? p : j, j1 = (nib & 2u) ? p + 1 : j0, j2 = (nib & 4u) ? p + 2 : j1, j3 = (nib & 8u) ? p + 3 : j2;
        out[q] = j0 | (j1 << 8) | (j2 << 16) | (j3 << 24);
        j = j3;
    }
}

template <bool GROUPED>
__device__ __forceinline__ void intra_tu_any(const HevcIntraArgs &a, const ffhip_hevc_tu &t, const int lane, int *s, int *s2,
                                             int *refbase, short *R, const ResPrefetch &rp, short *tile, const int wl,
                                             const bool tile_ok)
{
    switch (t.log2_size) {
    case 2: intra_tu<GROUPED, 2>(a, t, lane, s, s2, refbase, R, rp, tile, wl, tile_ok); break;
    case 3: intra_tu<GROUPED, 3>(a, t, lane, s, s2, refbase, R, rp, tile, wl, tile_ok); break;
    case 4: intra_tu<GROUPED, 4>(a, t, lane, s, s2, refbase, R, rp, tile, wl, tile_ok); break;
    default: intra_tu<GROUPED, 5>(a, t, lane, s, s2, refbase, R, rp, tile, wl, tile_ok); break;
    }
}

/* level-synchronous form: one launch per dependency level, one wave per TU */
__global__ __launch_bounds__(256) void k_hevc_intra(HevcIntraArgs a)
{
    __shared__ int nbA[4][NB_MAX], nbB[4][NB_MAX], refs[4][140];
    __shared__ __attribute__((aligned(16))) short resl[4][32 * 32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + w;
    if (item >= a.count) return;
    const ffhip_hevc_tu t = a.tus[a.work[item]];
    ResPrefetch rp;
    fetch_residual(a, t, lane, rp);
    intra_tu_any<false>(a, t, lane, nbA[w], nbB[w], refs[w], resl[w], rp, nullptr, 0, false);
}

/* Grouped form: ONE launch per picture.  The host cuts the TU list into groups -- the TUs whose
 * top-left corner falls into one window x window tile of one plane, contiguous in decode order
 * for any window no larger than the CTB -- and a wave takes groups by ticket and walks each in
 * list order.  Inside a group nothing goes through memory on the critical path: the wave keeps
 * the window's reconstructed samples in an LDS tile and reads its neighbours from there, has the
 * group's TU records in LDS (loaded 64 at a time) and fetches each residual block one TU ahead.
 * Hops between groups go through one done flag per TU.  Picture samples are stored, and
 * neighbours outside the window gathered, with agent-scope accesses (sc1: served by the
 * device-coherent level, never by a possibly stale per-XCD L2 line), so publishing a TU needs
 * only "my stores have completed" (an explicit s_waitcnt vmcnt(0)) before the flag store -- no
 * cache-wide write-back/invalidate.  Every TU a slot waits for lies in a group with a smaller ticket (the
 * host checks), whose wave is therefore already running or finished: no wave ever waits for
 * work that has not been picked up; the spin is bounded all the same and reports through the
 * pinned word ffhip_stream_sync looks at. */
#define SPIN_LIMIT (1 << 21)
#define SLOT_CHUNK 64

/* slot words -> wave-uniform registers.  q0 = first half of the TU record, q2 = the schedule's own words */
#define SGPR(v) ((unsigned)__builtin_amdgcn_readfirstlane((int)(v)))
__device__ __forceinline__ IntraSlot decode_slot(const u32x4 q0, const u32x4 q2)
{
    const unsigned d0 = SGPR(q0.x), d1 = SGPR(q0.y), d2 = SGPR(q0.z), d3 = SGPR(q0.w);
    const unsigned d8 = SGPR(q2.x), d9 = SGPR(q2.y), d10 = SGPR(q2.z), d11 = SGPR(q2.w);
    IntraSlot sl;
    sl.x = d0 & 0xffff; sl.y = d0 >> 16;
    sl.lg = d1 & 0xff; sl.cidx = (d1 >> 8) & 0xff; sl.mode = (d1 >> 16) & 0xff; sl.flags = d1 >> 24;
    sl.res_offset = d2; sl.res_scale = (int)d3;
    sl.wait_begin = d8; sl.wait_count = d9 & 0xff; sl.signal = (d9 >> 8) & 1; sl.tile_ok = (d9 >> 9) & 1;
    sl.tu_index = d10; sl.jt_off = d11;
    return sl;
}

/* A list the device planner refuses (groups that are not contiguous runs of the decode order for this window, more
 * than 64 TUs to wait for, an order its ticket rule cannot serve) is still decoded, correctly and slowly: ONE wave walks
 * it in decode order, which is always a valid order, every neighbour through memory.  A kernel of its own (launched
 * behind the grouped one, which returns at once in that case; this one returns at once in every other): inlined into the
 * grouped kernel it doubled that kernel's code, which is as large as the instruction cache as it is. */
__global__ __launch_bounds__(64) void k_hevc_intra_serial(HevcIntraArgs a)
{
    __shared__ short tile[TILE_CELLS];
    __shared__ int nbA[NB_MAX], nbB[NB_MAX];
    __shared__ __attribute__((aligned(16))) short resl[2][32 * 32];
    __shared__ __attribute__((aligned(16))) short resz[64];
    __shared__ u32x4 slots[SLOT_CHUNK * 3];
    const int lane = threadIdx.x;
    if (!a.plan_result) return;
    const uint32_t refused = a.plan_result[0] | (a.plan_result[2] > a.wait_cap ? 1u : 0u);
    if (!__builtin_amdgcn_readfirstlane((int)refused)) return;
    if (__builtin_amdgcn_readfirstlane((int)a.plan_result[6])) return; /* refused as INVALID (k_hevc_check_tus): nothing is written */
    resz[lane] = 0;
    wave_sync();
    HotArgs hot;
    hot.residual = (const __attribute__((address_space(1))) int16_t *)(unsigned long long)a.residual;
    hot.jt = (const __attribute__((address_space(1))) uint8_t *)(unsigned long long)a.jt;
    hot.wait_idx = nullptr;
    hot.bitdepth_y = a.bitdepth_y; hot.bitdepth_c = a.bitdepth_c;
    for (long long base = 0; base < a.n_tus; base += SLOT_CHUNK) {
        const int m = (int)(a.n_tus - base < SLOT_CHUNK ? a.n_tus - base : SLOT_CHUNK);
        for (int i = lane; i < 3 * m; i += 64) {
            const int k = i / 3, part = i - 3 * k;
            const ffhip_hevc_tu *tp = a.tus + base + k;
            const int c = tp->cidx;
            const uint32_t blk = (c == 0 ? a.jt_boff[0] : (c == 1 ? a.jt_boff[1] : a.jt_boff[2])) +
                                 (uint32_t)(tp->y >> 2) * (uint32_t)(c == 0 ? a.jt_bw[0] : (c == 1 ? a.jt_bw[1] : a.jt_bw[2])) + (uint32_t)(tp->x >> 2);
            u32x4 q = {0u, 0u, (uint32_t)(base + k), blk * JT_STRIDE}; /* no waits, no flag, no tile */
            if (part == 0) q = ((const u32x4 *)tp)[0];
            slots[i] = q;
        }
        wave_sync();
        for (int k = 0; k < m; k++) {
            IntraSlot cur = decode_slot(slots[3 * k], slots[3 * k + 2]);
            cur.tile_ok = 0; /* every neighbour through memory: the planner's answer assumes ONE run per window (for_each_dep), which a refused list need not have */
            ResPrefetch rp;
            JPrefetch jp;
            fetch_residual_g(hot, cur, lane, rp);
            fetch_jtable(hot, cur, lane, jp);
            GroupCtx sg = {};
            sg.cidx = (int)cur.cidx; sg.wl = 6; sg.wx0 = (int)(cur.x >> 6) << 6; sg.wy0 = (int)(cur.y >> 6) << 6;
            sg.stride = sg.cidx == 0 ? a.stride[0] : (sg.cidx == 1 ? a.stride[1] : a.stride[2]);
            sg.plane_rs = ffhip_rsrc(sg.cidx == 0 ? a.plane[0] : (sg.cidx == 1 ? a.plane[1] : a.plane[2]), 0xffffffffu);
            intra_tu_g_any(hot, sg, cur, lane, nbA, nbB, resl[k & 1], rp, jp, tile, resz, [] {});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the next TU may read these samples back from memory */
            wave_sync();
        }
        wave_sync();
    }
}

/* MINW: waves per SIMD the register allocation is asked to fit (2: the latency instance, 186 VGPRs; 3: the throughput instance for lists of many
 * groups, 168 VGPRs and a dozen spilled); CHUNK: slots of a group staged in LDS at a time (64 / 16: with the 128-byte zero block the throughput
 * instance's LDS comes to 14.4 KB, eleven one-wave workgroups per CU instead of eight) */
template <int MINW, int CHUNK>
__global__ __launch_bounds__(64, MINW) void k_hevc_intra_groups(HevcIntraArgs a)
{
    __shared__ short tile[TILE_CELLS];
    __shared__ int nbA[NB_MAX], nbB[NB_MAX];
    constexpr int RESL = MINW >= 3 ? 1 : 2; /* residual blocks staged in LDS: two, alternating, where the registers are plentiful; ONE in the throughput instance --
                                               a wave's LDS accesses are served in order, and 2 KB less is a twelfth wave per CU */
    __shared__ __attribute__((aligned(16))) short resl[RESL][32 * 32];
    __shared__ __attribute__((aligned(16))) short resz[64]; /* zeros: the residual of a TU without one (every pass reads the same 64) */
    __shared__ u32x4 slots[(CHUNK + 2) * 3];
    const int lane = threadIdx.x;
    uint32_t *flags = a.ctrl + CTRL_HDR;
    int n_groups = a.n_groups;
    if (lane == 0) {
        tile[TILE_CONST_Y] = (short)(1 << (a.bitdepth_y - 1));
        tile[TILE_CONST_C] = (short)(1 << (a.bitdepth_c - 1));
        tile[TILE_ZERO] = 0;
    }
    resz[lane] = 0;
    wave_sync();
    HotArgs hot;
    {
        unsigned long long p_res = (unsigned long long)a.residual, p_jt = (unsigned long long)a.jt, p_wait = (unsigned long long)a.wait_idx;
        int bdy = a.bitdepth_y, bdc = a.bitdepth_c;
        asm volatile("" : "+s"(p_res), "+s"(p_jt), "+s"(p_wait), "+s"(bdy), "+s"(bdc));
        hot.residual = (const __attribute__((address_space(1))) int16_t *)p_res;
        hot.jt = (const __attribute__((address_space(1))) uint8_t *)p_jt;
        hot.wait_idx = (const __attribute__((address_space(1))) uint32_t *)p_wait;
        hot.bitdepth_y = bdy; hot.bitdepth_c = bdc;
    }
    if (a.plan_result) {
        /* the schedule was built by the kernels in front of this one on the same stream; nobody on the host has looked
         * at it.  A list the device planner refuses is decoded by k_hevc_intra_serial, launched behind this kernel */
        const uint32_t refused = a.plan_result[0] | (a.plan_result[2] > a.wait_cap ? 1u : 0u);
        n_groups = (int)a.plan_result[1];
        if (__builtin_amdgcn_readfirstlane((int)refused)) return;
        /* The launch holds as many waves as the device can keep resident (eight per CU: registers); how many of them can be AT
         * WORK at once follows from the width of the list's dependency wavefront, which only the planner knows (a single 8K
         * picture: ~200 groups of one depth; a grid of 135 independent tiles: ~1 600; eight such pictures: more than the chip
         * holds).  A wave that holds a ticket far from its turn only polls, through the same memory path the working waves use
         * (one 8K picture on 1024 waves: 6.5 ms, on 256: 5.9; the 135-tile grid on 2048 / 1280 / 1024 / 640 waves: 2.36 / 1.68 /
         * 1.46 / 1.46 ms; eight of them: 8.7 / 9.4 / 10.0 / 11.9 ms -- tests/tools/bench_hevc_grid.py): half the width, at least
         * 256, is what those runs ask for; the waves beyond that leave now. */
        const uint32_t width = a.plan_result[4];
        /* two instances of this kernel are launched behind a device-built plan, and the plan's wavefront width says which one works: lists
         * with a thousand groups and more ready at once (grids of tiles) are bound by how many waves the chip holds -- the instance that fits
         * three waves per SIMD; a single picture's few hundred by one wave's latency -- the instance with all its registers */
        const bool wide = a.tp_width && width >= a.tp_width;
        if (wide != (MINW >= 3)) return;
        const uint32_t cap = ((width * a.width_pct) >> 7) + 16;
        if (width && blockIdx.x >= (cap > 256u ? cap : 256u)) return;
    }
    const __amdgpu_buffer_rsrc_t desc_rs = ffhip_rsrc(a.desc, 0xffffffffu), res_rs = ffhip_rsrc((const void *)hot.residual, PROG_NO_RESIDUAL);
    const int cell_lane4 = 2 * ((lane >> 2) * TILE_STRIDE + (lane & 3)), cell_lane8 = 2 * ((lane >> 3) * TILE_STRIDE + (lane & 7));
    /* the run of plain programs never touches the exec mask (a branch over a load makes the compiler's wait counts
     * pessimistic: it then waits for the loads it issued a moment ago): on a 4x4 TU lanes 16 .. 63 do what lane 0 does --
     * same words, same residual, same value to the same addresses */
    const int lane4 = lane < 16 ? lane : 0;
    const int cell_alias4 = 2 * ((lane4 >> 2) * TILE_STRIDE + (lane4 & 3));
    bool dead = false; /* a wave that gave up waiting (bounded spin): leaves through the loop heads, not from inside them */
    const unsigned shards = a.ticket_shards < gridDim.x ? (a.ticket_shards ? a.ticket_shards : 1u) : gridDim.x; /* every counter needs a wave of its own: one
                                                                                                                    that stays with it until it is used up */
    unsigned my_shard = blockIdx.x % shards;
    /* (Taking the NEXT ticket while the group at hand is still being worked on -- the ticket, the group record and the slots are three trips
     * to memory in a row between two groups, 4.2 us a group on the eight-picture grid, 15 % of the kernel's wave time
     * (tests/tools/diag_intra_trace_grid.py) -- was built and measured slower: grids of 1 / 4 / 8 pictures 1.19 / 2.40 / 4.07 ms against
     * 1.08 / 2.22 / 4.03, one 8K picture 5.30 against 5.18.  A ticket held by a busy wave is a ready group that waits for it while idle waves
     * hold later tickets.) */
    while (!dead) {
        unsigned ticket = 0;
        if (shards > 1) {
            /* Ticket t belongs to counter t mod shards (each in a line of its own); a wave starts at "its" counter and moves on when one is used
             * up.  With one counter the eight-picture grid's 207 000 tickets were 83 device-scope atomics a microsecond on one word: 4.00 ms
             * against 3.80 with four counters (four pictures 2.05 / 1.93).  The order argument holds per counter and across them: the smallest
             * unfinished ticket is either somebody's group at hand or not taken yet -- then a wave of its counter is at work on a smaller one,
             * which cannot be (FFHIP_HEVC_TICKET_SHARDS=1: the single counter). */
            for (unsigned tries = 0; tries < shards; tries++) {
                const unsigned sh = (my_shard + tries) % shards;
                if (lane == 0) ticket = __hip_atomic_fetch_add(&a.ctrl[a.ctrl_ticket + 32 * sh], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket) * shards + sh;
                if (ticket < (unsigned)n_groups) { my_shard = sh; break; }
            }
        } else {
            if (lane == 0) ticket = __hip_atomic_fetch_add(&a.ctrl[a.ctrl_ticket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
        }
        if (ticket >= (unsigned)n_groups) break;
        const u32x4 g = a.groups[ticket];
        GroupCtx gc;
        int plane_alias4 = 0, desc_alias4 = 0;
        gc.wl = (int)g.z;
#ifdef FFHIP_INTRA_TRACE
        if (a.trace && lane == 0) a.trace[12 * a.n_tus + ticket] = TRACE_NOW();
#endif
        for (unsigned base = 0; base < g.y && !dead; base += CHUNK) {
            const int m = (int)(g.y - base < CHUNK ? g.y - base : CHUNK);
            for (int i = lane; i < 3 * m; i += 64) slots[i] = a.sched[(size_t)(g.x + base) * 3 + i];
            if (lane == 0) { /* behind the last slot: sentinels (the fetch runs two TUs ahead) the run of plain programs below stops at */
                const u32x4 end = {PK_SLOW | PK_END, PROG_NO_RESIDUAL, 0u, 0u};
                slots[3 * m + 1] = end;
                slots[3 * m + 4] = end;
            }
            wave_sync();
            if (base == 0) { /* the group's plane and window, from its first TU */
                const unsigned d0 = SGPR(slots[0].x), d1 = SGPR(slots[0].y);
                gc.cidx = (int)((d1 >> 8) & 0xff);
                gc.wx0 = (int)(((d0 & 0xffff) >> gc.wl) << gc.wl);
                gc.wy0 = (int)(((d0 >> 16) >> gc.wl) << gc.wl);
                gc.stride = gc.cidx == 0 ? a.stride[0] : (gc.cidx == 1 ? a.stride[1] : a.stride[2]);
                gc.maxv = (1 << (gc.cidx == 0 ? hot.bitdepth_y : hot.bitdepth_c)) - 1;
                gc.plane_rs = ffhip_rsrc(gc.cidx == 0 ? a.plane[0] : (gc.cidx == 1 ? a.plane[1] : a.plane[2]), 0xffffffffu);
                const int dw = gc.cidx == 0 ? a.desc_w[0] : (gc.cidx == 1 ? a.desc_w[1] : a.desc_w[2]);
                gc.plane_lane[0] = 2 * ((lane >> 2) * gc.stride + (lane & 3));
                gc.plane_lane[1] = 2 * ((lane >> 3) * gc.stride + (lane & 7));
                gc.desc_lane[0] = 8 * ((lane >> 2) * dw + (lane & 3));
                gc.desc_lane[1] = 8 * ((lane >> 3) * dw + (lane & 7));
                plane_alias4 = 2 * ((lane4 >> 2) * gc.stride + (lane4 & 3));
                desc_alias4 = 8 * ((lane4 >> 2) * dw + (lane4 & 3));
            }
            /* Fetched TWO TUs ahead (an L2 hit takes longer than a small TU): the slot's program words, the pixel words
             * and the pixel's residual of EVERY TU (a generic TU's slot points them at nothing) -- straight-line code: a
             * taken branch costs a lone wave as much as eight instructions (tests/tools/microbench_lone_wave.hip).  Two
             * register sets, A for the TU at hand and B for the one behind it; the run of common TUs below is unrolled
             * by two so that they never move.  Fetched ONE TU ahead, behind one unlikely branch: what the rest needs -- a
             * generic TU's slot, residual block and substitution table, the flags to poll, the table of a program that
             * reads halo cells. */
            ProgSlot psA, psB;
            ProgPrefetch ppA, ppB;
            IntraSlot cur;
            ResPrefetch rp;
            JPrefetch jp;
            uint32_t widx = 0;
#define PREFETCH_PROGRAM(ps, pp, kk) do { \
                const u32x4 q1_ = slots[3 * (kk) + 1]; \
                ps.packed = SGPR(q1_.x); ps.res_off = SGPR(q1_.y); ps.desc_off = SGPR(q1_.z); ps.plane_off = SGPR(q1_.w); \
                const bool big_ = (ps.packed & PK_LG3) != 0; \
                const unsigned long long dd_ = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(desc_rs, big_ ? gc.desc_lane[1] : desc_alias4, (int)ps.desc_off, 0)); \
                pp.d.x = (unsigned)dd_; pp.d.y = (unsigned)(dd_ >> 32); \
                /* a TU without residual has res_off = PROG_NO_RESIDUAL, the end of the resource: the load returns 0 */ \
                pp.res = (short)__builtin_amdgcn_raw_buffer_load_b16(res_rs, 2 * (big_ ? lane : lane4) + (int)ps.res_off, 0, 0); \
                } while (0)
            /* for a TU that is not a plain program: its slot, and what it reads besides the pixel words */
            auto fetch_extras = [&](const ProgSlot &ps, const int j) {
                cur = decode_slot(slots[3 * j], slots[3 * j + 2]);
                if (PK_KIND(ps.packed) == PROG_GENERIC) {
                    fetch_jtable(hot, cur, lane, jp);
                    fetch_residual_g(hot, cur, lane, rp);
                } else if (ps.packed & PK_OUTSIDE) {
                    fetch_jtable(hot, cur, lane, jp);
                }
                if (cur.wait_count) widx = hot.wait_idx[cur.wait_begin + (lane < (int)cur.wait_count ? lane : 0)];
            };
#define PLAIN_PROGRAM(ps, pp, kk) do { \
                const bool big_ = (ps.packed & PK_LG3) != 0; \
                const unsigned a0_ = pp.d.x & 0xffffu, a1_ = pp.d.x >> 16, a2_ = pp.d.y & 0xffffu; \
                if (__builtin_expect((ps.packed & PK_FILTER) != 0, 0)) program_filter_step(tile, lane, pp.d.y >> 16); \
                int r0_ = LDS_U16(a0_), r1_ = LDS_U16(a1_); \
                int v_; \
                if (__builtin_expect(PK_KIND(ps.packed) == PROG_ANGULAR, 1)) { \
                    v_ = ((32 - (int)a2_) * r0_ + (int)a2_ * r1_ + 16) >> 5; \
                } else { /* the other kinds: rarer, behind one taken branch */ \
                    const unsigned a3_ = pp.d.y >> 16, kind_ = PK_KIND(ps.packed); \
                    const int lg_ = big_ ? 3 : 2, n_ = 1 << lg_, le_ = big_ ? lane : lane4; \
                    if (kind_ == PROG_PLANAR) { \
                        const int x_ = le_ & (n_ - 1), y_ = le_ >> lg_; \
                        int r2_, r3_; \
                        if (ps.packed & PK_FILTER) { /* LEFT(y), TOP(x), TOP(n), LEFT(n) among the smoothed cells */ \
                            r0_ = (int)((const unsigned short *)tile)[TILE_F + 15 - y_]; r1_ = (int)((const unsigned short *)tile)[TILE_F + 17 + x_]; \
                            r2_ = (int)((const unsigned short *)tile)[TILE_F + 25]; r3_ = (int)((const unsigned short *)tile)[TILE_F + 7]; \
                        } else { \
                            r2_ = LDS_U16(a2_); r3_ = LDS_U16(a3_); \
                        } \
                        v_ = ((n_ - 1 - x_) * r0_ + (x_ + 1) * r2_ + (n_ - 1 - y_) * r1_ + (y_ + 1) * r3_ + n_) >> (lg_ + 1); \
                    } else if (kind_ == PROG_DC) { \
                        const int dc_ = (__builtin_amdgcn_readfirstlane(row_sum16(LDS_U16(a2_))) + n_) >> (lg_ + 1); \
                        v_ = dc_; \
                        if (a3_ == 1) v_ = (r0_ + 2 * dc_ + r1_ + 2) >> 2; \
                        else if (a3_ == 2) v_ = (r1_ + 3 * dc_ + 2) >> 2; \
                        else if (a3_ == 3) v_ = (r0_ + 3 * dc_ + 2) >> 2; \
                    } else { \
                        const int r2_ = LDS_U16(a2_); \
                        const int vn_ = ((32 - (int)a2_) * r0_ + (int)a2_ * r1_ + 16) >> 5; \
                        const int ve_ = clip3i(0, (1 << hot.bitdepth_y) - 1, r0_ + ((r1_ - r2_) >> 1)); \
                        v_ = a3_ ? ve_ : vn_; \
                    } \
                } \
                const short rec_ = (short)clip3i(0, gc.maxv, (int)(short)(v_ & 0xffff) + (int)pp.res); \
                *(short *)((char *)tile + (ps.packed >> 16) + (big_ ? cell_lane8 : cell_alias4)) = rec_; \
                __builtin_amdgcn_raw_buffer_store_b16(rec_, gc.plane_rs, big_ ? gc.plane_lane[1] : plane_alias4, (int)SGPR(ps.plane_off), FFHIP_AUX_SC1); \
                if (__builtin_expect((ps.packed & PK_SIGNAL) != 0, 0)) { /* somebody outside the group reads this TU: publish it once its stores have completed */ \
                    const unsigned tu_ = SGPR(slots[3 * (kk) + 2].z); \
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
                    if (lane == 0 && (int)tu_ != a.debug_withhold) __hip_atomic_store(flags + tu_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                } \
                } while (0)
#define SWAP_SETS() do { const ProgSlot ts_ = psA; psA = psB; psB = ts_; const ProgPrefetch tp_ = ppA; ppA = ppB; ppB = tp_; } while (0)
#ifdef FFHIP_INTRA_TRACE
#define TRACE_TU_BEGIN(ps) const unsigned long long tr0 = TRACE_NOW(); unsigned long long tr1 = tr0; \
    const unsigned tr_kind = PK_KIND(ps.packed) | ((ps.packed & PK_OUTSIDE) ? 8u : 0u), tr_lg = PK_KIND(ps.packed) ? ((ps.packed & PK_LG3) ? 3u : 2u) : cur.lg
#define TRACE_TU_END() do { if (a.trace && lane == 0) { \
        unsigned long long *tr = a.trace + 12 * (size_t)(g.x + base + k); /* by schedule slot */ \
        tr[0] = tr0; tr[1] = tr1; tr[2] = TRACE_NOW(); \
        tr[3] = ((unsigned long long)(tr_kind | (tr_lg << 4)) << 56) | ((unsigned long long)ticket << 32) | ((unsigned long long)blockIdx.x << 12) | (unsigned)(base + k); \
        tr[4] = g_stamp[0]; tr[5] = g_stamp[1]; tr[6] = g_stamp[2]; tr[7] = g_stamp[3]; \
        tr[8] = g_stamp[4]; tr[9] = g_stamp[5]; tr[10] = g_stamp[6]; tr[11] = g_stamp[7]; } } while (0)
#else
#define TRACE_TU_BEGIN(ps) do { } while (0)
#define TRACE_TU_END() do { } while (0)
#endif
            int k = 0;
            PREFETCH_PROGRAM(psA, ppA, 0);
            PREFETCH_PROGRAM(psB, ppB, 1);
            bool have_extras = false; /* the extras of TU k (set A) are in cur / rp / jp / widx, or on their way */
            for (;;) {
                /* invariant: set A = TU k, set B = TU k + 1.
                 * The run of common TUs: angular programs with nothing to wait for, nobody to tell and no halo cell.
                 * It goes on while the TU at hand AND the one behind it are such (nothing but the two register sets is
                 * carried round the loop); the sentinels behind the last slot end it */
                while (__builtin_expect(((psA.packed | psB.packed) & PK_SLOW) == 0, 1)) {
                    {
                        TRACE_TU_BEGIN(psA);
                        PLAIN_PROGRAM(psA, ppA, k);
                        TRACE_TU_END();
                    }
                    ++k;
                    PREFETCH_PROGRAM(psA, ppA, k + 1);
                    {
                        TRACE_TU_BEGIN(psB);
                        PLAIN_PROGRAM(psB, ppB, k);
                        TRACE_TU_END();
                    }
                    ++k;
                    PREFETCH_PROGRAM(psB, ppB, k + 1);
                }
                if (!(psA.packed & PK_SLOW)) {
                    /* a plain TU in front of one that is not: the other's extras leave now, one TU ahead */
                    if (!(psB.packed & PK_END)) fetch_extras(psB, k + 1);
                    {
                        TRACE_TU_BEGIN(psA);
                        PLAIN_PROGRAM(psA, ppA, k);
                        TRACE_TU_END();
                    }
                    ++k;
                    PREFETCH_PROGRAM(psA, ppA, k + 1);
                    SWAP_SETS();
                    have_extras = true;
                }
                if (psA.packed & PK_END) break;
                if (!have_extras) fetch_extras(psA, k); /* only the first slot of a chunk comes here without them */
                do { /* TUs that are not plain programs, one after the other: the extras of each were fetched during the one before */
                    const ProgSlot &ps = psA;
                    const ProgPrefetch &pp = ppA;
                    TRACE_TU_BEGIN(ps);
                    const bool is_prog = PK_KIND(ps.packed) != PROG_GENERIC;
                    if (cur.wait_count) { /* wait for the TUs of other groups this one reads (at most 64 of them) */
                        const uint32_t *fp = flags + widx;
                        int spins = 0;
                        for (;;) {
                            const unsigned done = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (__builtin_amdgcn_ballot_w64(done == 0) == 0) break;
                            if (++spins > SPIN_LIMIT || ((spins & 15) == 0 && SGPR(__hip_atomic_load(&a.ctrl[a.ctrl_abort], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) {
                                if (lane == 0) {
                                    __hip_atomic_store(&a.ctrl[a.ctrl_abort], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    __hip_atomic_store(a.async_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                }
                                dead = true;
                                break;
                            }
#ifndef FFHIP_POLL_FAST
#define FFHIP_POLL_FAST 16
#endif
#ifndef FFHIP_POLL_SLOW_SLEEP
#define FFHIP_POLL_SLOW_SLEEP 16 /* ~0.5 us between polls of a wave far from its turn.  With 1024 waves in the launch this sent 15 M flag
                                    reads per 8K picture through the L2 the working waves use and 64 was better; with 256 waves 8 .. 32 are alike */
#endif
                            if (spins < FFHIP_POLL_FAST) __builtin_amdgcn_s_sleep(1);
                            else for (uint32_t r_ = 0; r_ < a.poll_reps; r_++) __builtin_amdgcn_s_sleep(FFHIP_POLL_SLOW_SLEEP); /* far from ready */
                        }
                        if (dead) break;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); /* ordering only: no cache-wide invalidate */
                    }
#ifdef FFHIP_INTRA_TRACE
                    tr1 = TRACE_NOW();
#endif
                    /* the extras of the NEXT TU, if it needs any, leave now and land in registers of their own while this TU
                     * runs; they move into place behind it, when they have long arrived.  (Fetched behind this TU into the
                     * registers the next one reads them from, the compiler's copies between the call sites of the fetch
                     * waited for them on the spot: a trip to memory between any two generic TUs.) */
                    const bool nx_have = (psB.packed & (PK_SLOW | PK_END)) == PK_SLOW;
                    u32x4 nq0 = {0u, 0u, 0u, 0u}, nq2 = {0u, 0u, 0u, 0u};
                    JPrefetch jpn;
                    ResPrefetch rpn;
                    uint32_t widxn = 0;
                    /* zeroed on purpose (twenty v_mov per generic TU): left undefined, the compiler lets the fetch below load straight
                     * into the registers the NEXT TU reads, and the copies at the tail then wait for the loads on the spot (measured:
                     * config-5 mix 5.56 -> 6.10 ms; round 6, "defined" by empty asm statements instead of moves -- no instruction at all --: the same, 4.88 -> 5.50) */
                    jpn.j[0] = jpn.j[1] = jpn.j[2] = 0; rpn.wide = false; rpn.v[0] = rpn.v[1] = nq0;
                    auto fetch_next_extras = [&]() {
                        if (nx_have) {
                            nq0 = slots[3 * (k + 1)]; nq2 = slots[3 * (k + 1) + 2];
                            const IntraSlot nx = decode_slot(nq0, nq2);
                            if (PK_KIND(psB.packed) == PROG_GENERIC) {
                                fetch_jtable(hot, nx, lane, jpn);
                                fetch_residual_g(hot, nx, lane, rpn);
                            } else if (psB.packed & PK_OUTSIDE) {
                                fetch_jtable(hot, nx, lane, jpn);
                            }
                            if (nx.wait_count) widxn = hot.wait_idx[nx.wait_begin + (lane < (int)nx.wait_count ? lane : 0)];
                        }
                    };
                    if (is_prog) {
                        STAMP(0);
                        if (ps.packed & PK_OUTSIDE) intra_program_halo(gc, (int)cur.x, (int)cur.y, 1 << cur.lg, lane, jp.j[0], tile);
                        STAMP(1); STAMP(2);
                        if (ps.packed & PK_LG3) intra_program<3>(hot, gc, ps, lane, pp, tile, cell_lane8);
                        else intra_program<2>(hot, gc, ps, lane, pp, tile, cell_lane4);
                        STAMP(3);
                        fetch_next_extras(); /* a program is short: behind it */
                    } else {
                        intra_tu_g_any(hot, gc, cur, lane, nbA, nbB, resl[k & (RESL - 1)], rp, jp, tile, resz, fetch_next_extras);
                    }
                    if (cur.signal) { /* somebody outside the group reads this TU: publish it once its stores have completed */
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* compiler ordering; no L2-wide write-back */
                        /* the fence alone lowers to s_waitcnt lgkmcnt(0): the flag must not overtake the sample stores
                         * (MI355X_MICROARCH.md: every storing wave drains its vector-memory counter before it signals).
                         * Sending the flag one TU later instead -- when the next residual, fetched behind these stores,
                         * has been consumed -- was measured no faster: the reader waits for the flag either way */
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0 && (int)cur.tu_index != a.debug_withhold)
                            __hip_atomic_store(flags + cur.tu_index, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (!is_prog) wave_sync(); /* the next TU reuses the neighbour scratch (a program touches only the tile, in order) */
                    TRACE_TU_END();
                    ++k;
                    PREFETCH_PROGRAM(psA, ppA, k + 1); /* set A: TU k + 1, set B: TU k */
                    SWAP_SETS();
                    STAMP(5);
                    have_extras = nx_have;
                    if (nx_have) { /* set A is the TU the early fetch was for */
                        cur = decode_slot(nq0, nq2);
                        jp = jpn; rp = rpn; widx = widxn;
                    }
                    STAMP(6);
                } while ((psA.packed & (PK_SLOW | PK_END)) == PK_SLOW);
                if (dead) break;
            }
            wave_sync(); /* slots[] is about to be overwritten */
        }
    }
}
#undef SGPR

/* ------------------------------------------------------------------------ host */

/* what the device planner said about this thread's last list (ffhip_debug_hevc_plan_result) */
static thread_local const uint32_t *g_last_plan_result = nullptr;
static thread_local hipStream_t g_last_plan_stream = nullptr;
extern "C" int ffhip_debug_hevc_plan_result(uint32_t out[8])
{
    if (!out || !g_last_plan_result) return FFHIP_EINVAL;
    FFHIP_CHECK(hipStreamSynchronize(g_last_plan_stream), FFHIP_EIO);
    FFHIP_CHECK(hipMemcpy(out, g_last_plan_result, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost), FFHIP_EIO);
    return FFHIP_OK;
}


#define SCRATCH_HEVC_INTRA 3

extern "C" size_t ffhip_hevc_plan_gpu_words(long long n_tus, const int pw[3], const int ph[3], const int wl[3]);
extern "C" int ffhip_hevc_plan_gpu(const ffhip_hevc_tu *d_tus, long long n_tus, const int pw[3], const int ph[3], const int wl[3],
                                   uint32_t *scratch, hipStream_t st, const u32x4 **sched, const u32x4 **groups, const uint32_t **wait_idx,
                                   int *n_groups, const uint32_t **d_result, uint32_t *wait_cap_out);

extern "C" int ffhip_hevc_plan_gpu_checked(const ffhip_hevc_tu *d_tus, long long n_tus, const int pw[3], const int ph[3], const int wl[3],
                                           uint32_t *scratch, hipStream_t st, const u32x4 **sched, const u32x4 **groups, const uint32_t **wait_idx,
                                           int *n_groups, const uint32_t **d_result, uint32_t *wait_cap_out, const int *check, int *async_err,
                                           const FfhipPlanHooks *hooks, uint32_t *also_zero, size_t also_zero_words);

struct GroupPlan {
    std::vector<u32x4> sched;  /* 3 per slot */
    std::vector<u32x4> groups;
    std::vector<uint32_t> wait;
};

/* Cut the (validated) list into window-tile groups in order of first appearance and collect, per
 * TU, the TUs of OTHER groups it reads, whether some other group reads it, and whether all its
 * available neighbours inside the window were written by its own group (then the kernel may take
 * them from its LDS tile).  Returns false when some TU would wait for a group with a larger
 * ticket (window larger than the coding tree block, or an exotic list): the caller then tries a
 * smaller window or falls back to the level-synchronous form. */
static bool plan_groups(const ffhip_hevc_tu *tus, long long n_tus, const int pw[3], const int ph[3], const int win_log2[3],
                        GroupPlan &out, const uint32_t jt_boff[3])
{
    /* scratch kept between calls: a picture's worth of maps is reallocated and refilled otherwise */
    static thread_local std::vector<int32_t> owner[3], gid_of[3];
    struct Meta { uint32_t group, wait_begin, slot; uint8_t wait_count, signal, tile_ok; };
    static thread_local std::vector<Meta> meta;
    static thread_local std::vector<uint32_t> gcount, gdepth, gfirst, order, gbase;
    int bw[3], gw[3];
    for (int c = 0; c < 3; c++) {
        bw[c] = (pw[c] + 3) / 4;
        gw[c] = pw[c] > 0 ? ((pw[c] - 1) >> win_log2[c]) + 1 : 0;
        owner[c].assign((size_t)bw[c] * (size_t)((ph[c] + 3) / 4), -1);
        gid_of[c].assign((size_t)gw[c] * (size_t)(ph[c] > 0 ? ((ph[c] - 1) >> win_log2[c]) + 1 : 0), -1);
    }
    const auto T0 = std::chrono::steady_clock::now();
    meta.resize((size_t)n_tus);
    gcount.clear();
    gfirst.clear();
    /* ---- pass 1 (sequential, light): groups in order of first appearance, slot inside the group, block owners ----
     * contiguous: every group is one run of the list; then a group is complete before a later one starts,
     * which is what makes the dependency depths of pass 3 final when they are read */
    bool contiguous = true;
    uint32_t cur_group = ~0u;
    for (long long i = 0; i < n_tus; i++) {
        const ffhip_hevc_tu &t = tus[i];
        const int c = t.cidx, n = 1 << t.log2_size, wl = win_log2[c];
        int32_t &gslot = gid_of[c][(size_t)(t.y >> wl) * gw[c] + (t.x >> wl)];
        if (gslot < 0) {
            gslot = (int32_t)gcount.size();
            gcount.push_back(0);
            gfirst.push_back((uint32_t)i);
        } else if ((uint32_t)gslot != cur_group) {
            contiguous = false;
        }
        cur_group = (uint32_t)gslot;
        Meta &m = meta[(size_t)i];
        m.group = cur_group; m.signal = 0; m.tile_ok = 1;
        m.slot = gcount[cur_group]++;
        int32_t *orow = owner[c].data() + (size_t)(t.y >> 2) * bw[c] + (t.x >> 2);
        for (int by = 0; by < n / 4; by++, orow += bw[c])
            for (int bx = 0; bx < n / 4; bx++) orow[bx] = (int32_t)i;
    }
    const auto T1 = std::chrono::steady_clock::now();
    /* ---- pass 2 (parallel over TU ranges): who reads whom.  The owner map is complete; a TU only
     * depends on TUs before it in the list (a block whose owner comes later held older content when
     * the sequential decoder looked at it) ---- */
    /* the scratch vectors are thread_local: worker threads must go through pointers taken here */
    Meta *const mp = meta.data();
    const int32_t *const ownp[3] = {owner[0].data(), owner[1].data(), owner[2].data()};
    const char *pt = FFHIP_ENV("FFHIP_PLAN_THREADS");
    /* one thread unless asked: on the 16-core share of an MI355X box 2-8 threads were no faster
     * (2.5-5.0 ms against 2.7 ms for this pass on 172k TUs: thread start-up and the shared maps eat the gain) */
    const int n_threads = pt ? std::max(1, std::min(16, atoi(pt))) : 1;
    std::vector<std::vector<uint32_t>> waits((size_t)n_threads);
    std::atomic<bool> bad(false);
    auto scan = [&](int th) {
        const long long lo = n_tus * th / n_threads, hi = n_tus * (th + 1) / n_threads;
        std::vector<uint32_t> &w = waits[(size_t)th];
        w.reserve((size_t)(hi - lo) * 2);
        for (long long i = lo; i < hi; i++) {
            const ffhip_hevc_tu &t = tus[i];
            const int c = t.cidx, n = 1 << t.log2_size, wl = win_log2[c];
            Meta &m = mp[(size_t)i];
            const uint32_t g = m.group;
            int32_t deps[72];
            int nd = 0;
            bool tile_ok = true;
            const int wx0 = (t.x >> wl) << wl, wy0 = (t.y >> wl) << wl, wsz = 1 << wl;
            const int32_t *own = ownp[c];
            auto dep = [&](int px, int py) {
                int32_t j = own[(size_t)(py >> 2) * bw[c] + (px >> 2)];
                if (j >= i) j = -1;
                const bool mine = j >= 0 && mp[(size_t)j].group == g;
                if (j >= 0 && !mine) {
                    bool dup = false;
                    for (int q = nd - 1; q >= 0 && !dup; q--) dup = deps[q] == j; /* neighbours repeat back to back */
                    if (!dup && nd < 72) deps[nd++] = j;
                }
                if (!mine && px >= wx0 && px < wx0 + wsz && py >= wy0 && py < wy0 + wsz) tile_ok = false; /* not in my LDS copy */
            };
            if (t.flags & 1) dep(t.x - 1, t.y - 1);
            for (int k = 0; k < 2 * n; k += 4) {
                if ((t.avail_top >> k) & 0xf) dep(t.x + k, t.y - 1);
                if ((t.avail_left >> k) & 0xf) dep(t.x - 1, t.y + k);
            }
            if (nd > 64) { bad = true; return; }
            m.tile_ok = tile_ok;
            m.wait_count = (uint8_t)nd;
            m.wait_begin = (uint32_t)w.size(); /* relative to this thread's list until pass 3 */
            for (int q = 0; q < nd; q++) {
                Meta &mj = mp[(size_t)deps[q]];
                if (mj.group > g) { bad = true; return; }
                __atomic_store_n(&mj.signal, (uint8_t)1, __ATOMIC_RELAXED);
                w.push_back((uint32_t)deps[q]);
            }
        }
    };
    if (n_threads == 1) scan(0);
    else {
        std::vector<std::thread> pool;
        for (int th = 1; th < n_threads; th++) pool.emplace_back(scan, th);
        scan(0);
        for (auto &th : pool) th.join();
    }
    if (bad) return false;
    const auto T2 = std::chrono::steady_clock::now();
    /* ---- pass 3 (sequential, light): one wait list, dependency depth per group ---- */
    size_t total_wait = 0;
    for (auto &w : waits) total_wait += w.size();
    out.wait.resize(std::max<size_t>(total_wait, 1));
    out.wait[0] = 0;
    gdepth.assign(gcount.size(), 0);
    {
        size_t base = 0;
        for (int th = 0; th < n_threads; th++) {
            const long long lo = n_tus * th / n_threads, hi = n_tus * (th + 1) / n_threads;
            const std::vector<uint32_t> &w = waits[(size_t)th];
            if (!w.empty()) memcpy(out.wait.data() + base, w.data(), w.size() * sizeof(uint32_t));
            for (long long i = lo; i < hi; i++) {
                Meta &m = meta[(size_t)i];
                uint32_t depth = gdepth[m.group];
                for (unsigned q = 0; q < m.wait_count; q++) depth = std::max(depth, gdepth[meta[w[m.wait_begin + q]].group] + 1);
                gdepth[m.group] = depth;
                m.wait_begin += (uint32_t)base;
            }
            base += w.size();
        }
    }
    const auto T3 = std::chrono::steady_clock::now();
    /* Tickets go out in dependency-depth order (ties: decode order), so the waves that hold tickets
     * are the ones near the ready front rather than thousands of groups ahead of it, polling.
     * Every group a group waits for has a smaller depth, hence a smaller ticket.  (Depths are only
     * trusted for contiguous groups; otherwise decode order, which pass 2 checked is valid.) */
    const size_t ng = gcount.size();
    order.resize(ng);
    for (size_t g = 0; g < ng; g++) order[g] = (uint32_t)g;
    if (contiguous && !FFHIP_ENV("FFHIP_HEVC_INTRA_DECODE_ORDER"))
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return gdepth[x] < gdepth[y]; });
    gbase.resize(ng);
    out.groups.resize(ng);
    uint32_t run = 0;
    for (size_t k = 0; k < ng; k++) {
        const uint32_t g = order[k];
        gbase[g] = run;
        u32x4 rec;
        rec.x = run;
        rec.y = gcount[g];
        rec.z = (uint32_t)win_log2[tus[gfirst[g]].cidx];
        rec.w = 0;
        out.groups[k] = rec;
        run += gcount[g];
    }
    out.sched.resize((size_t)n_tus * 3);
    static_assert(sizeof(ffhip_hevc_tu) == 32, "slot layout");
    const uint32_t *const gbasep = gbase.data();
    auto emit = [&](int th) {
        const long long lo = n_tus * th / n_threads, hi = n_tus * (th + 1) / n_threads;
        for (long long i = lo; i < hi; i++) {
            const Meta &m = mp[(size_t)i];
            u32x4 *q = &out.sched[(size_t)(gbasep[m.group] + m.slot) * 3];
            memcpy(q, &tus[i], 32);
            q[2].x = m.wait_begin;
            q[2].y = (uint32_t)m.wait_count | ((uint32_t)m.signal << 8) | ((uint32_t)m.tile_ok << 9);
            q[2].z = (uint32_t)i;
            q[2].w = (jt_boff[tus[i].cidx] + (uint32_t)(tus[i].y >> 2) * (uint32_t)bw[tus[i].cidx] + (uint32_t)(tus[i].x >> 2)) * JT_STRIDE;
        }
    };
    if (n_threads == 1) emit(0);
    else {
        std::vector<std::thread> pool;
        for (int th = 1; th < n_threads; th++) pool.emplace_back(emit, th);
        emit(0);
        for (auto &th : pool) th.join();
    }
    if (FFHIP_ENV("FFHIP_PLAN_TIMES")) {
        const auto T4 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "plan: setup+pass1 %ld us, pass2 %ld us, pass3 %ld us, order+emit %ld us (threads %d)\n", us(T0, T1), us(T1, T2), us(T2, T3), us(T3, T4), n_threads);
    }
    return true;
}

/* Host-side passes over a TU list (validation, the contiguity test) are ~2-4 ns per TU and thread: 7 ms for the 1.8 million TUs of
 * eight 8K grids, more than the device needs for them.  Lists of 2^17 TUs and more are cut into pieces for up to 16 threads
 * (started per call: ~20 us each, they work while the others start). */
template <class F>
static void host_parallel_for(long long n, F &&fn)
{
    const long long min_piece = 1 << 16;
    unsigned hw = std::thread::hardware_concurrency();
    long long nt = n / min_piece;
    nt = nt > 16 ? 16 : nt;
    nt = hw && nt > (long long)hw ? (long long)hw : nt;
    if (nt < 2) { fn(0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    const long long piece = (n + nt - 1) / nt;
    for (long long k = 1; k < nt; k++) th.emplace_back([&fn, k, piece, n]() { fn(k * piece, std::min(n, (k + 1) * piece)); });
    fn(0, std::min(n, piece));
    for (auto &t : th) t.join();
}

/* Are the groups of this window -- the TUs whose top-left corner falls into one window tile of one plane -- contiguous
 * runs of the list?  (Then a group is complete before a later one starts, the condition of the grouped kernel.)  Two rules in one
 * pass, a byte map per plane and rule; scratch kept per thread:
 *   bit 0  runs of the list AS IT IS: a TU opens a run where its window differs from that of the record in front of it
 *   bit 1  runs of every plane's OWN subsequence: ... from that of the previous record of the same plane.  This is the rule the
 *          reference's order needs: it decodes coding unit by coding unit, the unit's luma tree, then Cb, then Cr
 *          (coding/hevc.c:5013-5180 calling decode_intra_block :4665-4805), so a coding tree block with several coding units
 *          switches planes INSIDE every 64x64 area.  The planes do not read each other, so the device planner may work on the
 *          list sorted by plane (ffhip_hevc_plan_gpu.hip, k_part_*), where bit 1 is what bit 0 is here.
 * Bit 0 implies bit 1. */
/* sampled: only the records of every 64th stretch of 4096 are looked at -- every 256th from a million records on -- (large lists, whose full test runs on the device: a window that
 * is not contiguous there is refused by the planner and the list decoded by the serial kernel -- exact, slow, and only for a list whose
 * coding-tree-block size changes between the sampled stretches) */
#define SAMPLED_OUT(i) ((((i) >> 12) & (n_tus >= (1LL << 20) ? 255 : 63)) != 0)
static int groups_contiguous(const ffhip_hevc_tu *tus, long long n_tus, const int pw[3], const int ph[3], const int win_log2[3], const bool sampled = false)
{
    static thread_local std::vector<uint8_t> seen[3];
    int gw[3];
    size_t cnt[3];
    for (int c = 0; c < 3; c++) {
        gw[c] = pw[c] > 0 ? ((pw[c] - 1) >> win_log2[c]) + 1 : 0;
        cnt[c] = (size_t)gw[c] * (size_t)(ph[c] > 0 ? ((ph[c] - 1) >> win_log2[c]) + 1 : 0);
        seen[c].assign(2 * cnt[c], 0); /* [0, cnt): the list as it is; [cnt, 2 cnt): the plane's own subsequence */
    }
    uint8_t *const map[3] = {seen[0].data(), seen[1].data(), seen[2].data()};
    std::atomic<bool> twice_raw{false}, twice_plane{false};
    auto window_of = [&](const ffhip_hevc_tu &t) -> long long { /* -1: an unvalidated record of a sampled list (the device pass refuses it) */
        const int c = t.cidx;
        if (c > 2 || t.x >= pw[c] || t.y >= ph[c]) return -1;
        return (long long)(t.y >> win_log2[c]) * gw[c] + (t.x >> win_log2[c]);
    };
    /* a TU opens a run where its window differs from its predecessor's: stateless per TU under the first rule, so the list is cut into
     * pieces for as many threads as pay (a window entered by two pieces is entered twice all the same: the mark is an atomic exchange);
     * under the second rule a piece -- and a sampled stretch -- first looks back for the last record of each plane in front of it */
    host_parallel_for(sampled ? 1 : n_tus, [&](long long b, long long e) { /* (a sample is one thread's work: starting sixteen costs more than the pass) */
        if (sampled) e = n_tus;
        long long last[3] = {-2, -2, -2}; /* the window of the plane's previous record; -2 = not looked up yet */
        auto look_back = [&](long long i) {
            int missing = 3;
            last[0] = last[1] = last[2] = -1;
            for (long long j = i - 1; j >= 0 && j >= i - 4096 && missing; j--) { /* (further back than any coding tree block reaches: a run that old is taken for a new one) */
                const int c = tus[j].cidx;
                if (c > 2 || last[c] != -1) continue;
                const long long w = window_of(tus[j]);
                if (w < 0) continue;
                last[c] = w; missing--;
            }
        };
        look_back(b);
        for (long long i = b; i < e && !twice_plane.load(std::memory_order_relaxed); i++) {
            if (sampled && SAMPLED_OUT(i)) { i |= 4095; if (i + 1 < e) look_back(i + 1); continue; }
            const ffhip_hevc_tu &t = tus[i];
            const long long w = window_of(t);
            if (w < 0) continue;
            const int c = t.cidx;
            if (last[c] != w) {
                last[c] = w;
                if (__atomic_exchange_n(map[c] + cnt[c] + w, (uint8_t)1, __ATOMIC_RELAXED)) twice_plane.store(true, std::memory_order_relaxed);
            }
            if (i > 0 && tus[i - 1].cidx == c && window_of(tus[i - 1]) == w) continue;
            if (!twice_raw.load(std::memory_order_relaxed) && __atomic_exchange_n(map[c] + w, (uint8_t)1, __ATOMIC_RELAXED)) twice_raw.store(true, std::memory_order_relaxed);
        }
    });
    const bool plane_ok = !twice_plane.load();
    return (plane_ok && !twice_raw.load() ? 1 : 0) | (plane_ok ? 2 : 0);
}

/* The list sorted by plane (stable), on the host: for the host planner and ffhip_hevc_intra_plan, what k_part_* do for the device planner.
 * perm[k] = the caller's index of sorted record k. */
static void sort_by_plane(const ffhip_hevc_tu *tus, long long n_tus, std::vector<ffhip_hevc_tu> &sorted, std::vector<uint32_t> *perm)
{
    size_t cnt[3] = {0, 0, 0};
    for (long long i = 0; i < n_tus; i++) cnt[tus[i].cidx > 2 ? 2 : tus[i].cidx]++;
    size_t at[3] = {0, cnt[0], cnt[0] + cnt[1]};
    sorted.resize((size_t)n_tus);
    if (perm) perm->resize((size_t)n_tus);
    for (long long i = 0; i < n_tus; i++) {
        const size_t k = at[tus[i].cidx > 2 ? 2 : tus[i].cidx]++;
        sorted[k] = tus[i];
        if (perm) (*perm)[k] = (uint32_t)i;
    }
}
/* the largest luma window (from `wl` down to 8x8) under which the list's groups are contiguous runs, by the plane's own subsequence; *by_plane:
 * NOT by the list as it is, i.e. the list has to be sorted by plane for that window.  0 when there is none.  FFHIP_HEVC_BY_PLANE=0 keeps to the
 * list as it is (the rule until round 5), =1 sorts whenever the sort alone does not make the window smaller (tests: both forms on every list). */
static int pick_window(const ffhip_hevc_tu *tus, long long n_tus, const int pw[3], const int ph[3], int wl, const bool sampled, bool *by_plane)
{
    const char *bp = FFHIP_ENV("FFHIP_HEVC_BY_PLANE");
    const int force = bp ? atoi(bp) : -1;
    const int cs = (pw[1] > 0 && pw[1] * 2 <= pw[0] + 1) ? 1 : 0;
    wl = wl < 3 ? 3 : (wl > 6 ? 6 : wl);
    for (; wl >= 3; wl--) {
        const int win[3] = {wl, wl - cs, wl - cs};
        const int bits = groups_contiguous(tus, n_tus, pw, ph, win, sampled);
        if (force == 0 ? (bits & 1) : (bits & 2)) {
            *by_plane = force == 0 ? false : (force == 1 ? true : !(bits & 1));
            return wl;
        }
    }
    *by_plane = false;
    return 0;
}

/* the window search both entry points share: the requested (or default) luma window, halved until a
 * plan exists; chroma windows cover the same picture area */
static bool plan_with_window_search(const ffhip_hevc_tu *tus, long long n_tus, const int pw[3], const int ph[3], int wl,
                                    GroupPlan &plan, int *used_wl, const uint32_t jt_boff[3])
{
    wl = wl < 3 ? 3 : (wl > 6 ? 6 : wl);
    const int cs = (pw[1] > 0 && pw[1] * 2 <= pw[0] + 1) ? 1 : 0;
    for (; wl >= 3; wl--) {
        const int win[3] = {wl, wl - cs, wl - cs};
        if (plan_groups(tus, n_tus, pw, ph, win, plan, jt_boff)) {
            if (used_wl) *used_wl = wl;
            return true;
        }
    }
    return false;
}

/* Host only (no device needed): the schedule ffhip_hevc_intra_recon would build for a VALIDATED list.
 * out_ticket[i] = ticket of TU i's group, out_wait[i] = number of TUs of other groups it waits for
 * (either may be NULL); stats = {groups, luma window log2 used, wait entries, TUs that may use the LDS tile}.
 * Returns FFHIP_EINVAL when no window gives a deadlock-free ticket order (the caller would use levels). */
extern "C" int ffhip_hevc_intra_plan(const ffhip_hevc_tu *h_tus, long long n_tus, int width_y, int height_y, int width_c,
                                     int height_c, int window_log2, uint32_t *out_ticket, uint32_t *out_wait, int32_t *stats)
{
    if (!h_tus || n_tus <= 0 || width_y <= 0 || height_y <= 0) return FFHIP_EINVAL;
    const int pw[3] = {width_y, width_c, width_c}, ph[3] = {height_y, height_c, height_c};
    GroupPlan plan;
    int wl = 0;
    const uint32_t no_table[3] = {0, 0, 0};
    /* as ffhip_hevc_intra_recon does: a list that interleaves the planes inside a window is planned sorted by plane */
    std::vector<ffhip_hevc_tu> sorted;
    std::vector<uint32_t> perm;
    bool by_plane = false;
    const int want = window_log2 ? window_log2 : FFHIP_HEVC_INTRA_WINDOW_LOG2;
    (void)pick_window(h_tus, n_tus, pw, ph, want, false, &by_plane);
    if (by_plane) sort_by_plane(h_tus, n_tus, sorted, &perm);
    if (!plan_with_window_search(by_plane ? sorted.data() : h_tus, n_tus, pw, ph, want, plan, &wl, no_table)) return FFHIP_EINVAL;
    int tile_ok = 0;
    for (size_t g = 0; g < plan.groups.size(); g++)
        for (uint32_t k = 0; k < plan.groups[g].y; k++) {
            const u32x4 q = plan.sched[(size_t)(plan.groups[g].x + k) * 3 + 2];
            const uint32_t i = by_plane ? perm[q.z] : q.z;
            if (out_ticket) out_ticket[i] = (uint32_t)g;
            if (out_wait) out_wait[i] = q.y & 0xff;
            tile_ok += (q.y >> 9) & 1;
        }
    if (stats) { stats[0] = (int32_t)plan.groups.size(); stats[1] = wl; stats[2] = (int32_t)plan.wait.size(); stats[3] = tile_ok; }
    return FFHIP_OK;
}

/* ffhip_hevc_intra_recon_tiles runs a call's chunks as a pipeline: a chunk's pre-pass (validation, planner, substitution table, programs) on
 * `plan` -- a stream of the library's own, one pre-pass behind the other -- and its grouped kernel on `groups` (the caller's stream and a second
 * library stream in turn), behind `plan_done`.  Scratch of its own per chunk (`scratch_kind`); the substitution table and the per-pixel words,
 * which are indexed by position in the planes, are shared (`jt_desc`: the chunks cover different areas). */
struct IntraRoles {
    hipStream_t plan, groups;
    hipEvent_t plan_done;
    int scratch_kind;
    uint32_t *jt_desc;
    bool big_call; /* the call this chunk is cut from is a large list (2^17 records and more): the host samples, the device validates, as for the whole list */
};
static int intra_recon_impl(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus,
                            const int16_t *d_residual, int16_t *d_y, int16_t *d_cb, int16_t *d_cr,
                            int width_y, int height_y, int y_stride, int width_c, int height_c,
                            int uv_stride, int bitdepth_y, int bitdepth_c, void *stream, const IntraRoles *roles)
{
    if (n_tus < 0 || n_tus > 0x7fffffffLL) return FFHIP_EINVAL;
    if (n_tus == 0) return FFHIP_OK;
    if (!h_tus || !d_tus || !d_y || width_y <= 0 || height_y <= 0 || y_stride < width_y) return FFHIP_EINVAL;
    if (bitdepth_y < 8 || bitdepth_y > 15 || bitdepth_c < 8 || bitdepth_c > 15) return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    const int pw[3] = {width_y, width_c, width_c}, ph[3] = {height_y, height_c, height_c};
    /* validation: field ranges, the block inside its plane, and no availability bit pointing outside the plane */
    /* lists of 2^17 TUs and more: the host looks at a sample (a bad record there is refused here, at once), every record is checked by a
     * kernel in front of the planner (k_hevc_check_tus), which refuses the call through the stream -- or by a full host pass below, should the
     * call not take the device planner */
    const bool big_list = (n_tus >= (1LL << 17) || (roles && roles->big_call)) && !FFHIP_ENV("FFHIP_HEVC_HOST_CHECK");
    std::atomic<bool> bad{false}, any_res{false};
    auto validate = [&](const bool sampled) {
    host_parallel_for(sampled ? 1 : n_tus, [&](long long b, long long e) {
        if (sampled) e = n_tus;
        bool res = false, ok = true;
        for (long long i = b; i < e && ok; i++) {
            if (sampled && SAMPLED_OUT(i)) { i |= 4095; continue; }
            const ffhip_hevc_tu &t = h_tus[i];
            const int c = t.cidx, n = 1 << t.log2_size;
            if (c > 2 || t.log2_size < 2 || t.log2_size > 5 || t.pred_mode > 34) { ok = false; break; }
            if (c > 0 && (!d_cb || !d_cr || uv_stride < width_c)) { ok = false; break; }
            if (t.x + n > pw[c] || t.y + n > ph[c]) { ok = false; break; }
            const unsigned long long span = n == 32 ? ~0ull : (1ull << (2 * n)) - 1;
            const unsigned long long top = t.avail_top & span, left = t.avail_left & span;
            const int room_x = pw[c] - t.x, room_y = ph[c] - t.y; /* samples that exist right of x0 / below y0 */
            if ((top || (t.flags & 1)) && t.y == 0) ok = false;
            if ((left || (t.flags & 1)) && t.x == 0) ok = false;
            if (room_x < 64 && (top >> room_x)) ok = false;
            if (room_y < 64 && (left >> room_y)) ok = false;
            res |= (t.flags & 2) != 0;
        }
        if (!ok) bad.store(true, std::memory_order_relaxed);
        if (res) any_res.store(true, std::memory_order_relaxed);
    });
    };
    const bool host_times = FFHIP_ENV("FFHIP_PLAN_TIMES") != nullptr;
    const auto TH0 = std::chrono::steady_clock::now();
    validate(big_list);
    const auto TH1 = std::chrono::steady_clock::now();
    if (bad.load()) return FFHIP_EINVAL;
    bool has_res = any_res.load();
    bool fully_validated = !big_list;
    auto validate_fully = [&]() -> bool { /* before anything on the host walks the whole list */
        if (!fully_validated) { validate(false); fully_validated = true; has_res = any_res.load(); }
        return !bad.load() && !(has_res && !d_residual);
    };
    /* wavefront levels at 4x4-block granularity, per plane: only the level-synchronous form needs them */
    std::vector<std::vector<uint32_t>> lists;
    auto build_levels = [&]() {
        std::vector<int> lvl[3];
        int bw[3];
        for (int c = 0; c < 3; c++) {
            bw[c] = (pw[c] + 3) / 4;
            lvl[c].assign((size_t)(c == 0 || (d_cb && d_cr) ? bw[c] * ((ph[c] + 3) / 4) : 0), -1);
        }
        for (long long i = 0; i < n_tus; i++) {
            const ffhip_hevc_tu &t = h_tus[i];
            const int c = t.cidx, n = 1 << t.log2_size;
            int lv = 0;
            auto dep = [&](int px, int py) { lv = std::max(lv, lvl[c][(size_t)(py / 4) * bw[c] + px / 4] + 1); };
            if (t.flags & 1) dep(t.x - 1, t.y - 1);
            for (int k = 0; k < 2 * n; k++) {
                if ((t.avail_top >> k) & 1) dep(t.x + k, t.y - 1);
                if ((t.avail_left >> k) & 1) dep(t.x - 1, t.y + k);
            }
            for (int by = t.y / 4; by < (t.y + n) / 4; by++)
                for (int bx = t.x / 4; bx < (t.x + n) / 4; bx++) lvl[c][(size_t)by * bw[c] + bx] = lv;
            if ((size_t)lv >= lists.size()) lists.resize((size_t)lv + 1);
            lists[(size_t)lv].push_back((uint32_t)i);
        }
    };
    if (has_res && !d_residual) return FFHIP_EINVAL;
    hipStream_t st = roles ? roles->plan : (hipStream_t)stream; /* everything in front of the grouped kernel */
    const int scratch_kind = roles ? roles->scratch_kind : SCRATCH_HEVC_INTRA;
    HevcIntraArgs a = {};
    /* the records the schedule's TU indices refer to: the caller's, or -- for a list that interleaves the planes inside a scheduling window --
     * a copy sorted by plane in the call's scratch (pick_window) */
    const ffhip_hevc_tu *list = d_tus;
    a.tus = d_tus; a.residual = d_residual;
    a.plane[0] = d_y; a.plane[1] = d_cb; a.plane[2] = d_cr;
    a.stride[0] = y_stride; a.stride[1] = uv_stride; a.stride[2] = uv_stride;
    a.bitdepth_y = bitdepth_y; a.bitdepth_c = bitdepth_c;
    {   /* drives the bounded-spin give-up path in tests: the waiters of that TU run into SPIN_LIMIT and report FFHIP_EIO */
        const char *dw = FFHIP_ENV("FFHIP_DEBUG_WITHHOLD_TU");
        a.debug_withhold = dw ? atoi(dw) : -1;
    }

    /* grouped single-launch form unless FFHIP_HEVC_INTRA_MODE=levels (diagnostics) or no window works */
    const char *mode_env = FFHIP_ENV("FFHIP_HEVC_INTRA_MODE");
    const bool want_groups = !(mode_env && !strcmp(mode_env, "levels"));
    int *async_err = want_groups ? ffhip_async_err_word() : nullptr;
    const bool offsets_fit = (long long)y_stride * height_y < (1LL << 30) && (long long)uv_stride * (height_c > 0 ? height_c : 1) < (1LL << 30);
    /* geometry of the substitution table (and of the device planner's owner map): 4x4 blocks of the planes in use */
    const int pwc[3] = {pw[0], (d_cb && d_cr) ? pw[1] : 0, (d_cb && d_cr) ? pw[2] : 0};
    size_t jt_blocks = 0;
    JTabArgs ja = {};
    for (int c = 0; c < 3; c++) {
        ja.bw[c] = (pwc[c] + 3) / 4;
        ja.boff[c] = (uint32_t)jt_blocks;
        if (pwc[c] > 0) jt_blocks += (size_t)ja.bw[c] * (size_t)((ph[c] + 3) / 4);
        a.jt_bw[c] = ja.bw[c];
        a.jt_boff[c] = ja.boff[c];
    }
    const size_t w_jt = (jt_blocks * JT_STRIDE + 256 + 3) / 4; /* padded: the kernel fetches 64 / 192 entries per TU whatever its size */
    /* the per-pixel programs: 8 bytes per sample of the planes in use */
    ProgArgs pa = {};
    size_t desc_px = 0;
    for (int c = 0; c < 3; c++) {
        pa.desc_w[c] = pwc[c];
        pa.desc_off[c] = (uint32_t)desc_px;
        if (pwc[c] > 0) desc_px += (size_t)pwc[c] * (size_t)ph[c];
        a.desc_w[c] = pa.desc_w[c];
        a.desc_off[c] = pa.desc_off[c];
    }
    const size_t w_desc = desc_px * 2 + 2;
    auto enqueue_programs = [&](uint32_t *words, const int win[3], size_t n_slots, hipStream_t ps) {
        pa.sched = (u32x4 *)a.sched; pa.n_slots = (uint32_t)n_slots; pa.jt = a.jt;
        pa.records = list;
        pa.desc = (uint2 *)(((uintptr_t)words + 7) & ~(uintptr_t)7);
        for (int c = 0; c < 3; c++) { pa.wl[c] = win[c]; pa.stride[c] = a.stride[c]; pa.jt_bw[c] = ja.bw[c]; pa.jt_boff[c] = ja.boff[c]; }
        pa.plan_result = a.plan_result; pa.wait_cap = a.wait_cap;
        a.desc = pa.desc;
        hipLaunchKernelGGL(k_hevc_intra_program, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, ps, pa);
    };
    /* The substitution table depends on the TU list alone: for large lists it is built on the calling thread's side stream, NEXT TO the planner's
     * kernels (which are chains of dependent loads with the chip mostly idle), and joined in front of the first kernel that reads it. */
    bool jt_forked = false;
    FfhipSide side = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto enqueue_jtable = [&](uint32_t *words, bool may_fork) -> int {
        ja.tus = d_tus; ja.n = (uint32_t)n_tus; ja.jt = (uint8_t *)words;
        a.jt = ja.jt;
        hipStream_t js = st;
        if (may_fork && n_tus >= (1 << 15) && !FFHIP_ENV("FFHIP_HEVC_JT_INLINE") && ffhip_side_stream_get(&side) == FFHIP_OK) {
            FFHIP_CHECK(hipEventRecord((hipEvent_t)side.fork, st), FFHIP_EIO); /* behind whatever of an earlier call still reads the table's memory */
            FFHIP_CHECK(hipStreamWaitEvent((hipStream_t)side.stream, (hipEvent_t)side.fork, 0), FFHIP_EIO);
            js = (hipStream_t)side.stream;
            jt_forked = true;
        }
        hipLaunchKernelGGL(k_hevc_intra_jtable, dim3((unsigned)((n_tus + 255) / 256)), dim3(256), 0, js, ja);
        if (jt_forked) FFHIP_CHECK(hipEventRecord((hipEvent_t)side.join, js), FFHIP_EIO);
        return FFHIP_OK;
    };
    auto join_jtable = [&]() -> int { /* in front of the first kernel on `st` that reads the table */
        if (jt_forked) FFHIP_CHECK(hipStreamWaitEvent(st, (hipEvent_t)side.join, 0), FFHIP_EIO);
        jt_forked = false;
        return FFHIP_OK;
    };
    if (want_groups && async_err && offsets_fit && jt_blocks * JT_STRIDE < (1ull << 32) && desc_px < (1ull << 29) /* 32-bit byte offsets into a plane, the table and the pixel words */) {
        const char *we = FFHIP_ENV("FFHIP_HEVC_INTRA_WINDOW");
        const char *wv = FFHIP_ENV("FFHIP_HEVC_INTRA_WAVES");
        /* device-planned launches start as many waves as can be resident and trim themselves to the planner's wavefront width
         * (k_hevc_intra_groups); host-planned ones keep the flat cap */
        const size_t max_waves = wv ? (size_t)std::max(1, atoi(wv)) : FFHIP_HEVC_INTRA_WAVES;
        const size_t resident_waves = wv ? max_waves : (size_t)std::max(FFHIP_HEVC_INTRA_WAVES, ffhip_resident_waves((const void *)k_hevc_intra_groups<2, 64>, 64));
        const size_t resident_waves_tp = wv ? max_waves : (size_t)std::max(FFHIP_HEVC_INTRA_WAVES, ffhip_resident_waves((const void *)k_hevc_intra_groups<3, 16>, 64));
        /* The throughput instance (three waves per SIMD, 168 VGPRs) takes plans whose widest wavefront is 3 000 groups and more -- grids of several
         * pictures: four / eight pictures 1.86 / 3.58 ms against 1.98 / 3.85 with the latency instance; one picture's 135 tiles (width 1 652) are
         * indifferent (0.78 / 0.77).  While every ticket was an atomic on ONE word in a cache line that waiting waves polled, the extra waves bought
         * nothing (4.28 against 4.29 ms at eight pictures) and the instance was off.  FFHIP_HEVC_INTRA_TP_WIDTH=<width> moves the threshold, 0 = never. */
        { const char *tw = FFHIP_ENV("FFHIP_HEVC_INTRA_TP_WIDTH"); a.tp_width = tw ? (uint32_t)std::max(0, atoi(tw)) : 3000u; }
        { const char *pr = FFHIP_ENV("FFHIP_HEVC_INTRA_POLL_REPS"); a.poll_reps = pr ? (uint32_t)std::max(1, atoi(pr)) : 1u; }
        { const char *tsx = FFHIP_ENV("FFHIP_HEVC_TICKET_SHARDS"); a.ticket_shards = tsx ? (uint32_t)std::min(8, std::max(1, atoi(tsx))) : 4u; }
        { const char *wp = FFHIP_ENV("FFHIP_HEVC_INTRA_WIDTH_PCT"); a.width_pct = wp ? (uint32_t)std::max(1, atoi(wp)) : 128u; }
        /* the schedule: built on the device (ffhip_hevc_plan_gpu.hip) unless FFHIP_HEVC_PLAN=host; lists whose groups are
         * not contiguous runs of the decode order come back from there and take the host planner with its window search */
        const char *pe = FFHIP_ENV("FFHIP_HEVC_PLAN");
        /* a list whose groups are not contiguous runs of the decode order even at the smallest window would be refused by the device
         * planner and decoded by ONE wave (k_hevc_intra_serial: exact, seconds for an 8K list): such a list takes the host planner
         * with its window search -- or the levels form -- right away; the serial kernel stays for what only the device can find
         * (more than 64 TUs to wait for, an order its ticket rule cannot serve) */
        bool by_plane = false;
        const int dev_cs = (pw[1] > 0 && pw[1] * 2 <= pw[0] + 1) ? 1 : 0;
        int dev_wl = pick_window(h_tus, n_tus, pw, ph, we ? atoi(we) : 6, big_list, &by_plane);
        bool dev_ok = dev_wl != 0;
        if (!dev_ok) dev_wl = 3;
        const auto TH2 = std::chrono::steady_clock::now();
        if (pe && !strcmp(pe, "device")) dev_ok = true; /* tests: force the device planner (and with it the serial path of a list it refuses) */
        if (!(pe && !strcmp(pe, "host")) && dev_ok) {
            /* NOTHING below waits for the device: the schedule is enqueued, the grouped kernel is enqueued behind it and
             * reads the planner's verdict itself (a refused list takes its serial path).  The window is chosen here, on
             * the host, from the list alone: the largest one (up to the requested) whose groups are contiguous runs of
             * the decode order -- a 16x16 coding tree block stream needs 16, and finding that out on the device would
             * cost a round trip. */
            /* default window: 64x64, one group per coding tree block.  A group start costs a chain of dependent loads
             * (ticket, group record, slots, wait list, flags: ~3 us) and every window edge makes halo TUs; with the small
             * TUs running as per-pixel programs the longer serial walk through a 64x64 block costs less than that
             * (8K random quadtree: 14.8 ms against 17.3 with 32x32; the config-5 mix 8.1 against 12.0; 1080p and smaller
             * pictures are indifferent -- tests/tools/bench_intra_c5.py, bench_intra_sizes.py) */
            const int wl = dev_wl, cs = dev_cs;
            const int win[3] = {wl, wl - cs, wl - cs};
            const size_t w_plan = ffhip_hevc_plan_gpu_words(n_tus, pwc, ph, win), w_ctrl = CTRL_HDR + (size_t)n_tus;
            const bool shared_jt = roles && roles->jt_desc;
            uint32_t *g_work = ffhip_scratch(scratch_kind, stream, w_plan + w_ctrl + 16 + (shared_jt ? 0 : w_jt + w_desc));
            if (!g_work) return FFHIP_ENOMEM;
            uint32_t *const jt_words = shared_jt ? roles->jt_desc : g_work + ((w_plan + 3) & ~(size_t)3) + ((w_ctrl + 3) & ~(size_t)3);
            int n_groups = 0;
            /* the substitution table starts (on the side stream) behind the list's validation: a bad record's position would send its stores anywhere */
            /* ... and the per-pixel programs follow it there, behind k_plan_count (whose flags and wait counts are all they need of the schedule),
             * next to the ticket kernels and k_plan_emit on `stream`; the grouped kernel waits for both.  (Started right behind the validation
             * instead, from the TU records alone, with k_plan_emit settling the slot words afterwards -- the programs read no table any more --
             * they ran next to k_plan_owner and k_plan_count, which then took 341 and 383 us instead of 135 and 223: the eight-picture grid's
             * pre-pass 1.14 ms instead of 1.00.) */
            /* The side stream takes, behind the table: the depth sweep of the planner's cells (behind k_plan_owner, NEXT TO k_plan_count: three
             * workgroups that walk diagonals for 50 - 130 us), then -- behind k_plan_count -- the ticket kernels and k_plan_emit.  The per-pixel
             * programs, the one large kernel of that stretch, follow k_plan_count on `stream` itself, from what that kernel left (TU record,
             * flag byte, wait count: the slots' other quarters are being written next door).  (Until late in round 4 the programs were the side
             * stream's and the sweep ran next to them on `stream`: 360 us instead of 130, the long pole of the pre-pass.) */
            bool programs_forked = false, tickets_aside = false;
            auto ticket_stream = [&]() -> void * {
                if (!jt_forked || FFHIP_ENV("FFHIP_HEVC_SWEEP_INLINE") || FFHIP_ENV("FFHIP_HEVC_PROGRAMS_INLINE")) return nullptr;
                if (hipEventRecord((hipEvent_t)side.fork, st) != hipSuccess || hipStreamWaitEvent((hipStream_t)side.stream, (hipEvent_t)side.fork, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    return nullptr;
                }
                tickets_aside = true;
                return side.stream;
            };
            auto programs_early = [&](const uint8_t *flags, const uint32_t *wcount, const uint32_t *result) -> int {
                if (!jt_forked) return FFHIP_OK; /* a small list: everything on `stream`, in order */
                pa.tus = list; pa.flags = flags; pa.wcount = wcount; pa.refused = nullptr;
                a.sched = (const u32x4 *)g_work; /* where the planner puts the slots (ffhip_hevc_plan_gpu's layout starts with them) */
                a.plan_result = result; a.wait_cap = (uint32_t)(8 * (size_t)n_tus);
                if (tickets_aside) { /* on `stream`, right behind k_plan_count -- and behind the point the ticket kernels wait for */
                    FFHIP_CHECK(hipEventRecord((hipEvent_t)side.mid, st), FFHIP_EIO);
                    enqueue_programs(jt_words + w_jt, win, (size_t)n_tus, st);
                } else { /* the sweep and the ticket kernels stay on `stream`: the programs go next to them */
                    FFHIP_CHECK(hipEventRecord((hipEvent_t)side.mid, st), FFHIP_EIO);
                    FFHIP_CHECK(hipStreamWaitEvent((hipStream_t)side.stream, (hipEvent_t)side.mid, 0), FFHIP_EIO);
                    enqueue_programs(jt_words + w_jt, win, (size_t)n_tus, (hipStream_t)side.stream);
                    FFHIP_CHECK(hipEventRecord((hipEvent_t)side.join, (hipStream_t)side.stream), FFHIP_EIO);
                }
                programs_forked = true;
                return FFHIP_OK;
            };
            auto tickets_wait = [&]() -> int {
                FFHIP_CHECK(hipStreamWaitEvent((hipStream_t)side.stream, (hipEvent_t)side.mid, 0), FFHIP_EIO); /* recorded behind k_plan_count, in front of the programs */
                return FFHIP_OK;
            };
            auto tickets_enqueued = [&]() -> int {
                FFHIP_CHECK(hipEventRecord((hipEvent_t)side.join, (hipStream_t)side.stream), FFHIP_EIO);
                return FFHIP_OK;
            };
            struct Hook { decltype(enqueue_jtable) *fn; decltype(programs_early) *pe; uint32_t *words; JTabArgs *ja; decltype(ticket_stream) *ts; decltype(tickets_wait) *tw;
                          decltype(tickets_enqueued) *te; } hook = {&enqueue_jtable, &programs_early, jt_words, &ja, &ticket_stream, &tickets_wait, &tickets_enqueued};
            FfhipPlanHooks hooks = {};
            hooks.ctx = &hook;
            hooks.by_plane = by_plane ? 1 : 0;
            hooks.tus_used = &list;
            hooks.after_check = [](void *ctx, const unsigned *refused) -> int {
                Hook *h = (Hook *)ctx;
                h->ja->refused = refused;
                return (*h->fn)(h->words, true);
            };
            hooks.ticket_stream = [](void *ctx) -> void * { return (*((Hook *)ctx)->ts)(); };
            hooks.tickets_wait = [](void *ctx) -> int { return (*((Hook *)ctx)->tw)(); };
            hooks.tickets_enqueued = [](void *ctx) -> int { return (*((Hook *)ctx)->te)(); };
            if (!FFHIP_ENV("FFHIP_HEVC_PROGRAMS_INLINE"))
                hooks.after_count = [](void *ctx, const unsigned char *flags, const unsigned *wcount, const unsigned *result) -> int {
                    return (*((Hook *)ctx)->pe)(flags, wcount, result);
                };
            const int check[2] = {(d_cb && d_cr && uv_stride >= width_c) ? 1 : 0, d_residual ? 1 : 0};
            const int prc = ffhip_hevc_plan_gpu_checked(d_tus, n_tus, pwc, ph, win, g_work, st, &a.sched, &a.groups, &a.wait_idx, &n_groups, &a.plan_result, &a.wait_cap,
                                                        big_list ? check : nullptr, async_err, &hooks,
                                                        g_work + ((w_plan + 3) & ~(size_t)3), w_ctrl /* the ticket counter and the done flags: cleared by the planner's first launch */);
            if (prc < 0) { (void)join_jtable(); return prc; } /* (`stream` must not run ahead of the side stream's read of the caller's list) */
            a.ctrl = g_work + ((w_plan + 3) & ~(size_t)3);
            a.ctrl_ticket = (uint32_t)((32 - (((uintptr_t)a.ctrl >> 2) & 31)) & 31); a.ctrl_abort = a.ctrl_ticket + 32 * 9;
            a.async_err = async_err;
            a.n_groups = 0;
            a.n_tus = n_tus;
            a.tus = list;
            { const int jrc = join_jtable(); if (jrc) return jrc; } /* (the side stream's last record: behind the programs when they went there) */
            if (!programs_forked) { pa.tus = nullptr; pa.flags = nullptr; pa.wcount = nullptr; pa.refused = nullptr; enqueue_programs(jt_words + w_jt, win, (size_t)n_tus, st); }
#ifdef FFHIP_INTRA_TRACE
            a.trace = g_intra_trace;
#endif
            hipStream_t gst = st;
            if (roles) { /* a chunk of a pipelined call: the grouped kernel runs elsewhere, behind this pre-pass */
                FFHIP_CHECK(hipEventRecord(roles->plan_done, st), FFHIP_EIO);
                FFHIP_CHECK(hipStreamWaitEvent(roles->groups, roles->plan_done, 0), FFHIP_EIO);
                gst = roles->groups;
            }
            /* a chunk of a pipelined call leaves a share of the wave slots free: its waves hold ALL of their SIMD's registers (3 x 168), so with the
             * chip full of them the next chunk's pre-pass did not start before they left (its first kernel: 8 us alone, 550 us there), and no two
             * chunks' grouped kernels ever overlapped.  Tickets, not residency, are what a launch of any size is safe by. */
            size_t share = 100;
            if (roles && roles->jt_desc) { const char *sp = FFHIP_ENV("FFHIP_HEVC_TILE_WAVES_PCT"); share = sp ? (size_t)std::max(10, std::min(100, atoi(sp))) : 67; }
            const size_t rw = std::max<size_t>(64, resident_waves * share / 100), rw_tp = std::max<size_t>(64, resident_waves_tp * share / 100);
            hipLaunchKernelGGL((k_hevc_intra_groups<2, 64>), dim3((unsigned)std::min<size_t>((size_t)n_tus, rw)), dim3(64), 0, gst, a);
            /* (an instance with four waves per SIMD's worth of registers -- 128, 27 of them spilled -- would be thirteen waves per CU by LDS: measured,
             * 3.46 against 3.18 ms at eight pictures) */
            if (a.tp_width) hipLaunchKernelGGL((k_hevc_intra_groups<3, 16>), dim3((unsigned)std::min<size_t>((size_t)n_tus, rw_tp)), dim3(64), 0, gst, a);
            hipLaunchKernelGGL(k_hevc_intra_serial, dim3(1), dim3(64), 0, gst, a); /* does something only for a list the planner refused */
            g_last_plan_result = a.plan_result; g_last_plan_stream = gst;
            if (host_times) {
                const auto TH3 = std::chrono::steady_clock::now();
                auto us = [](auto x, auto y) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(y - x).count(); };
                fprintf(stderr, "intra_recon host: validate %ld us, window %ld us, enqueue %ld us (%lld TUs)\n", us(TH0, TH1), us(TH1, TH2), us(TH2, TH3), n_tus);
            }
            FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
            return FFHIP_OK;
        }
        if (!validate_fully()) return FFHIP_EINVAL; /* the host planner walks every record */
        if (roles) { /* (a chunk of a pipelined call that does not take the device planner: in line on its groups stream, which has the inputs) */
            FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
            st = roles->groups;
        }
        GroupPlan plan;
        int host_wl = 0;
        /* the host planner works on the list sorted by plane where the device planner would (pick_window): its slots' TU indices then refer to
         * the sorted records, which are uploaded next to the schedule */
        std::vector<ffhip_hevc_tu> h_sorted;
        if (big_list) (void)pick_window(h_tus, n_tus, pw, ph, we ? atoi(we) : FFHIP_HEVC_INTRA_WINDOW_LOG2, false, &by_plane); /* (the first answer came from a sample) */
        if (by_plane) sort_by_plane(h_tus, n_tus, h_sorted, nullptr);
        if (plan_with_window_search(by_plane ? h_sorted.data() : h_tus, n_tus, pw, ph, we ? atoi(we) : FFHIP_HEVC_INTRA_WINDOW_LOG2, plan, &host_wl, ja.boff)) {
            /* device image: sched | groups | wait | ctrl[CTRL_HDR] + one done flag per TU */
            const size_t w_sched = plan.sched.size() * 4, w_groups = plan.groups.size() * 4, w_wait = plan.wait.size();
            const size_t w_ctrl = CTRL_HDR + (size_t)n_tus;
            const size_t o_groups = w_sched, o_wait = o_groups + w_groups, o_ctrl = (o_wait + w_wait + 3) & ~(size_t)3;
            const size_t w_sorted = by_plane ? 8 * (size_t)n_tus + 16 : 0;
            FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO); /* the work buffer may still be in use by an earlier call */
            uint32_t *g_work = ffhip_scratch(scratch_kind, stream, o_ctrl + w_ctrl + 4 + w_jt + w_desc + w_sorted);
            if (!g_work) return FFHIP_ENOMEM;
            uint32_t *const jt_words = g_work + ((o_ctrl + w_ctrl + 3) & ~(size_t)3);
            if (by_plane) {
                uint32_t *ps = jt_words + w_jt + w_desc + 4;
                ps += (8 - (((uintptr_t)ps >> 2) & 7)) & 7;
                FFHIP_CHECK(hipMemcpy(ps, h_sorted.data(), (size_t)n_tus * sizeof(ffhip_hevc_tu), hipMemcpyHostToDevice), FFHIP_EIO);
                list = (const ffhip_hevc_tu *)ps;
                a.tus = list;
            }
            { const int jrc = enqueue_jtable(jt_words, false); if (jrc) return jrc; }
            FFHIP_CHECK(hipMemcpy(g_work, plan.sched.data(), w_sched * 4, hipMemcpyHostToDevice), FFHIP_EIO);
            FFHIP_CHECK(hipMemcpy(g_work + o_groups, plan.groups.data(), w_groups * 4, hipMemcpyHostToDevice), FFHIP_EIO);
            FFHIP_CHECK(hipMemcpy(g_work + o_wait, plan.wait.data(), w_wait * 4, hipMemcpyHostToDevice), FFHIP_EIO);
            FFHIP_CHECK(hipMemsetAsync(g_work + o_ctrl, 0, w_ctrl * 4, st), FFHIP_EIO);
            a.sched = (const u32x4 *)g_work;
            a.groups = (const u32x4 *)(g_work + o_groups);
            a.wait_idx = g_work + o_wait;
            a.ctrl = g_work + o_ctrl;
            g_last_plan_result = nullptr;
            a.ctrl_ticket = (uint32_t)((32 - (((uintptr_t)a.ctrl >> 2) & 31)) & 31); a.ctrl_abort = a.ctrl_ticket + 32 * 9;
            a.async_err = async_err;
            a.n_groups = (int)plan.groups.size();
            {
                const int cs = (pw[1] > 0 && pw[1] * 2 <= pw[0] + 1) ? 1 : 0;
                const int win[3] = {host_wl, host_wl - cs, host_wl - cs};
                enqueue_programs(jt_words + w_jt, win, (size_t)n_tus, st);
            }
            const unsigned wgs = (unsigned)std::min<size_t>(plan.groups.size(), max_waves); /* one wave each; waves loop over tickets */
            a.tp_width = 0;
            hipLaunchKernelGGL((k_hevc_intra_groups<2, 64>), dim3(wgs), dim3(64), 0, st, a);
            FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
            return FFHIP_OK;
        }
    }

    if (!validate_fully()) return FFHIP_EINVAL;
    if (roles && st == roles->plan) {
        FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
        st = roles->groups;
    }
    build_levels();
    std::vector<uint32_t> flat;
    flat.reserve((size_t)n_tus);
    for (auto &l : lists) flat.insert(flat.end(), l.begin(), l.end());
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
    uint32_t *g_work = ffhip_scratch(scratch_kind, stream, (size_t)n_tus);
    if (!g_work) return FFHIP_ENOMEM;
    FFHIP_CHECK(hipMemcpy(g_work, flat.data(), flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice), FFHIP_EIO);
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

extern "C" int ffhip_hevc_intra_recon(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus,
                                      const int16_t *d_residual, int16_t *d_y, int16_t *d_cb, int16_t *d_cr,
                                      int width_y, int height_y, int y_stride, int width_c, int height_c,
                                      int uv_stride, int bitdepth_y, int bitdepth_c, void *stream)
{
    return intra_recon_impl(h_tus, d_tus, n_tus, d_residual, d_y, d_cb, d_cr, width_y, height_y, y_stride, width_c, height_c, uv_stride, bitdepth_y, bitdepth_c, stream, nullptr);
}

/* The tile loop of a HEIF grid (format/heif.c:297-309: decode one tile after the other, no dependency between tiles) as ONE call over the
 * concatenated lists of independent pictures that share one plane set -- and as a PIPELINE: the list is cut at tile boundaries into up to four
 * chunks of about equal size; the pre-pass of chunk k + 1 (validation, planner, substitution table, per-pixel programs: a third of an
 * eight-picture call's span when it ran in front of the one grouped kernel) runs on streams of the library's own while chunk k reconstructs, and
 * the chunks' grouped kernels run on two streams in turn so that one chunk's start fills the other's tail.  What makes that legal is the
 * caller's word that tiles never reference each other; the results are those of ffhip_hevc_intra_recon on the whole list. */
#define SCRATCH_HEVC_TILES_JT 20
#define SCRATCH_HEVC_TILES_CHUNK 21 /* .. + 3 */
#define SCRATCH_HEVC_TILES_ONE 25   /* .. + 1 */
/* Who used a one-chunk scratch last: the scratch belongs to (kind, CALLER's stream), so its guard does too -- the parity of the call, and per scratch an
 * event recorded on the caller's stream behind the call's grouped kernel.  (Kept per calling thread until round 6: a thread that alternated between two
 * streams, or two threads on one stream, could let a pre-pass rewrite a schedule the grouped kernel of another call was still reading.)  Calls that
 * share a stream take turns for the length of their enqueue: they share the scratch. */
namespace {
struct TileGuard {
    std::mutex mu;
    unsigned parity = 0;
    bool recorded[2] = {false, false};
    hipEvent_t ev[2] = {nullptr, nullptr};
};
std::mutex g_tile_guard_mu;
std::map<void *, std::unique_ptr<TileGuard>> g_tile_guards;
TileGuard *tile_guard_for(void *stream)
{
    std::lock_guard<std::mutex> l(g_tile_guard_mu);
    std::unique_ptr<TileGuard> &g = g_tile_guards[stream];
    if (!g) {
        std::unique_ptr<TileGuard> n(new (std::nothrow) TileGuard);
        if (!n) return nullptr;
        if (hipEventCreateWithFlags(&n->ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&n->ev[1], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (n->ev[0]) (void)hipEventDestroy(n->ev[0]);
            return nullptr; /* (the map keeps an empty slot: the next call tries again) */
        }
        g = std::move(n);
    }
    return g.get();
}
} // namespace
extern "C" void ffhip_hevc_tiles_release(void) /* ffhip_release_caches: nothing of the library's is in flight, the scratches go as well */
{
    std::lock_guard<std::mutex> l(g_tile_guard_mu);
    for (auto &e : g_tile_guards)
        if (e.second) { (void)hipEventDestroy(e.second->ev[0]); (void)hipEventDestroy(e.second->ev[1]); }
    g_tile_guards.clear();
}
extern "C" int ffhip_hevc_intra_recon_tiles(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus, const long long *tile_first, int n_tiles,
                                            const int16_t *d_residual, int16_t *d_y, int16_t *d_cb, int16_t *d_cr, int width_y, int height_y, int y_stride,
                                            int width_c, int height_c, int uv_stride, int bitdepth_y, int bitdepth_c, void *stream)
{
    if (n_tus < 0 || n_tus > 0x7fffffffLL || n_tiles < 0 || (n_tiles > 0 && !tile_first)) return FFHIP_EINVAL;
    if (n_tus == 0) return FFHIP_OK;
    for (int k = 0; k < n_tiles; k++)
        if (tile_first[k] < (k ? tile_first[k - 1] : 0) || tile_first[k] > n_tus) return FFHIP_EINVAL;
    if (n_tiles > 0 && tile_first[0] != 0) return FFHIP_EINVAL;
    /* chunks (FFHIP_HEVC_TILE_CHUNKS=2..4; default 1: see below), cut at the tile boundaries nearest to equal shares */
    const char *ce = FFHIP_ENV("FFHIP_HEVC_TILE_CHUNKS");
    int want = ce ? std::max(1, std::min(4, atoi(ce))) : 1;
    long long cut[5] = {0, n_tus, n_tus, n_tus, n_tus};
    int chunks = 1;
    if (n_tiles > 1 && want > 1) {
        int ti = 1;
        for (int c = 1; c < want; c++) {
            const long long target = n_tus * c / want;
            while (ti < n_tiles && tile_first[ti] < target) ti++;
            if (ti >= n_tiles) break;
            /* the boundary at or behind the target, or the one in front of it when that is nearer */
            long long at = tile_first[ti];
            if (ti > 1 && target - tile_first[ti - 1] < at - target && tile_first[ti - 1] > cut[chunks - 1]) at = tile_first[ti - 1];
            if (at <= cut[chunks - 1] || at >= n_tus) continue;
            cut[chunks++] = at;
        }
        cut[chunks] = n_tus;
    }
    FfhipPipe pipe;
    const char *early_e = FFHIP_ENV("FFHIP_HEVC_TILE_EARLY");
    const bool early = !(early_e && early_e[0] == '0');
    if ((chunks == 1 && !early) || !ffhip_have_device() || ffhip_pipe_streams_get(&pipe) != FFHIP_OK)
        return intra_recon_impl(h_tus, d_tus, n_tus, d_residual, d_y, d_cb, d_cr, width_y, height_y, y_stride, width_c, height_c, uv_stride, bitdepth_y, bitdepth_c, stream, nullptr);
    /* The library's pre-pass stream may not touch a pipeline scratch (schedule, tables) before the grouped kernel of the call that used it last has
     * finished with it: an event recorded behind every call, waited for in front of the next call that takes the same scratch.  The one-chunk form
     * alternates between TWO scratches, so the pre-pass of call n + 1 only waits for call n - 1 and runs next to the tail of call n's grouped kernel
     * (whose waves leave as the wavefront narrows), the colour conversion behind it and the next residual batches. */
    if (chunks == 1) {
        TileGuard *const guard = tile_guard_for(stream);
        if (!guard) return FFHIP_EIO;
        std::lock_guard<std::mutex> turn(guard->mu);
        /* ONE chunk -- the default: cutting the list does not pay (below) --, but the pre-pass does not wait for `stream`: it reads the TU list alone,
         * so it runs while the stream is still busy with what the caller enqueued in front of this call -- the residual batches of this picture, the
         * colour conversion of the picture before.  (d_tus must be COMPLETE when the call is made: see the header.) */
        const char *db = FFHIP_ENV("FFHIP_HEVC_TILE_SCRATCHES");
        const unsigned par = (db && db[0] == '1') ? 0u : (guard->parity++ & 1u);
        IntraRoles roles;
        roles.plan = (hipStream_t)pipe.plan; roles.groups = (hipStream_t)stream; roles.plan_done = (hipEvent_t)pipe.ev[2];
        roles.scratch_kind = SCRATCH_HEVC_TILES_ONE + (int)par; roles.jt_desc = nullptr; roles.big_call = n_tus >= (1LL << 17);
        /* (with one scratch: the stream's call before; with two: the one before that) */
        if (guard->recorded[par]) FFHIP_CHECK(hipStreamWaitEvent(roles.plan, guard->ev[par], 0), FFHIP_EIO);
        const int rc1 = intra_recon_impl(h_tus, d_tus, n_tus, d_residual, d_y, d_cb, d_cr, width_y, height_y, y_stride, width_c, height_c, uv_stride, bitdepth_y, bitdepth_c, stream,
                                         &roles);
        if (hipEventRecord(guard->ev[par], (hipStream_t)stream) == hipSuccess) guard->recorded[par] = true;
        else { (void)hipGetLastError(); (void)hipStreamSynchronize((hipStream_t)stream); guard->recorded[par] = false; }
        return rc1;
    }
    if (!h_tus || !d_tus || !d_y || width_y <= 0 || height_y <= 0) return FFHIP_EINVAL;
    /* the tables indexed by position in the planes, shared by the chunks: the substitution table (JT_STRIDE bytes per 4x4 block) and the per-pixel
     * program words (8 bytes per sample) -- laid out as intra_recon_impl lays them out behind its own scratch */
    const bool chroma = d_cb && d_cr;
    size_t jt_blocks = 0, desc_px = 0;
    for (int c = 0; c < 3; c++) {
        const int w = c == 0 ? width_y : (chroma ? width_c : 0), h = c == 0 ? height_y : height_c;
        if (w > 0) { jt_blocks += (size_t)((w + 3) / 4) * (size_t)((h + 3) / 4); desc_px += (size_t)w * (size_t)h; }
    }
    const size_t w_jt = (jt_blocks * JT_STRIDE + 256 + 3) / 4, w_desc = desc_px * 2 + 2;
    uint32_t *jt_desc = ffhip_scratch(SCRATCH_HEVC_TILES_JT, stream, w_jt + w_desc + 16);
    if (!jt_desc) return FFHIP_ENOMEM;
    hipStream_t st = (hipStream_t)stream, s_plan = (hipStream_t)pipe.plan, s_g2 = (hipStream_t)pipe.groups2;
    hipEvent_t ev_in = (hipEvent_t)pipe.ev[0], ev_g2 = (hipEvent_t)pipe.ev[1];
    /* (measured on the eight-picture grid, profiles/r5_hevc_tiles_timeline_8x4.txt: a quarter of the tiles is a grouped kernel of 0.6 ms -- the length
     * of a tile's dependency chain -- where the whole list's is 1.73, and the grouped kernel's waves hold all of their SIMDs' registers, so the next
     * chunk's pre-pass does not start before they leave: 3.5 ms against 2.6 for the one launch; with two chunks, or with a third or half of the wave
     * slots left free, 2.65 - 3.8.  The cut stays as a tested switch; the default is one chunk.) */
    /* the library's streams start behind what the caller's holds (the residuals) */
    FFHIP_CHECK(hipEventRecord(ev_in, st), FFHIP_EIO);
    FFHIP_CHECK(hipStreamWaitEvent(s_plan, ev_in, 0), FFHIP_EIO);
    FFHIP_CHECK(hipStreamWaitEvent(s_g2, ev_in, 0), FFHIP_EIO);
    int rc = FFHIP_OK;
    for (int k = 0; k < chunks && rc == FFHIP_OK; k++) {
        IntraRoles roles;
        roles.plan = s_plan;
        roles.groups = (k & 1) ? s_g2 : st;
        roles.plan_done = (hipEvent_t)pipe.ev[2 + k];
        roles.scratch_kind = SCRATCH_HEVC_TILES_CHUNK + k;
        roles.jt_desc = jt_desc;
        roles.big_call = n_tus >= (1LL << 17);
        rc = intra_recon_impl(h_tus + cut[k], d_tus + cut[k], cut[k + 1] - cut[k], d_residual, d_y, d_cb, d_cr, width_y, height_y, y_stride, width_c, height_c, uv_stride,
                              bitdepth_y, bitdepth_c, stream, &roles);
    }
    /* the caller's stream continues behind everything, whatever happened */
    if (hipEventRecord(ev_g2, s_g2) != hipSuccess || hipStreamWaitEvent(st, ev_g2, 0) != hipSuccess) return FFHIP_EIO;
    if (hipEventRecord(ev_in, s_plan) != hipSuccess || hipStreamWaitEvent(st, ev_in, 0) != hipSuccess) return FFHIP_EIO;
    /* (the chunked form's streams start behind `stream`, which the call before has joined everything into) */
    return rc;
}

/* ffhip_hevc_intra_recon_tiles + YUV420_to_BGRA32_16bit (utils/colorspace.c:628-669; hevc.c:7260-7270) of the plane set as ONE call: the tile loop
 * of format/heif.c:297-309 with its colour conversion.  The colour kernel follows the grouped kernel on `stream` -- next to the pre-pass of the NEXT
 * call, which runs on the library's stream.  (The BGRA written by the grouped kernel itself -- the last group to finish inside a 64x64 cell, counted
 * over the three planes against the planner's runs per cell, converts the cell -- was built, bit-exact on every test of this entry point, and is
 * SLOWER: tests/tools/experiments/r5_hevc_fused_colour.patch.  A cell's conversion is a dozen dependent trips to memory on ONE wave of a kernel
 * whose throughput is its waves' latency: eight 8K pictures 3.37 ms against 2.89 with the colour kernel, four 1.69 against 1.51, one 0.64 against
 * 0.62.  The colour kernel moves the same bytes at the HBM's rate in 0.31 ms.) */
extern "C" int ffhip_yuv420_to_bgra_16(uint8_t *d_bgra, int pitch, const int16_t *d_y, const int16_t *d_u, const int16_t *d_v, int y_stride, int uv_stride, int ctbrows,
                                       int ctbcols, int ctbsize, int n_images, int64_t plane_stride_y, int64_t plane_stride_uv, int64_t image_stride, void *stream);
extern "C" int ffhip_hevc_decode_tiles(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus, const long long *tile_first, int n_tiles,
                                       const int16_t *d_residual, int16_t *d_y, int16_t *d_cb, int16_t *d_cr, int width_y, int height_y, int y_stride,
                                       int width_c, int height_c, int uv_stride, int bitdepth_y, int bitdepth_c, uint8_t *d_bgra, int64_t pitch, void *stream)
{
    if (!d_bgra || !d_cb || !d_cr || width_y <= 0 || height_y <= 0 || (width_y & 3) || (height_y & 1) || width_c != width_y / 2 || height_c != height_y / 2) return FFHIP_EINVAL;
    if (pitch < 4LL * width_y || (pitch & 15) || ((uintptr_t)d_bgra & 15) || pitch > 0x7fffffffLL) return FFHIP_EINVAL;
    const int rc = ffhip_hevc_intra_recon_tiles(h_tus, d_tus, n_tus, tile_first, n_tiles, d_residual, d_y, d_cb, d_cr, width_y, height_y, y_stride, width_c, height_c, uv_stride,
                                                bitdepth_y, bitdepth_c, stream);
    if (rc) return rc;
    return ffhip_yuv420_to_bgra_16(d_bgra, (int)pitch, d_y, d_cb, d_cr, y_stride, uv_stride, height_y / 2, width_y / 2, 2, 1, 0, 0, 0, stream);
}
