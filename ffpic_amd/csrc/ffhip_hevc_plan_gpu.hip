/*
 * ffhip_hevc_plan_gpu.hip -- the group schedule of ffhip_hevc_intra_recon built ON the device.
 *
 * The host planner (plan_groups in ffhip_hevc_intra.hip) costs ~30 ns per TU on one core: more than the kernel it
 * feeds once a picture has a few hundred thousand TUs (an 8K picture, a HEIF grid).  Everything it derives is a pure
 * function of the TU list, one TU at a time, given the map "which TU owns this 4x4 block":
 *   k_plan_owner   every TU stamps its index on its blocks, and says whether it starts a new RUN (a maximal stretch
 *                  of consecutive TUs whose top-left corners fall into the same window of the same plane)
 *   (scan)         run id per TU
 *   k_plan_count   per TU: the TUs of OTHER runs among its available neighbours (only earlier ones count: a block
 *                  stamped by a later TU held older content when the sequential decoder looked), whether all its
 *                  in-window neighbours are its own run's (LDS tile allowed), who must publish a done flag; per run
 *                  start: its position, and the claim on its window -- a window claimed by two runs means the list
 *                  is not "groups contiguous in decode order", and the caller falls back to the host planner
 *   (scan)         first wait entry per TU
 *   k_plan_emit    wait lists, schedule slots (position = TU index: runs in decode order ARE the ticket order, and a
 *                  TU an earlier-listed TU of another run precedes lies in a run that started earlier, i.e. has a
 *                  smaller ticket -- the deadlock-freedom condition of the grouped kernel), group records
 * Same output layout as the host planner; tickets in decode order (the host's dependency-depth order is a polling
 * optimisation worth ~3 %).
 */
#include "ffhip_internal.h"

#include <hipcub/hipcub.hpp>

struct PlanArgs {
    const ffhip_hevc_tu *tus;
    uint32_t n;
    int pw[3], ph[3], bw[3], gw[3], wl[3];
    uint32_t owner_off[3], win_off[3]; /* per plane: start inside owner[] / win_run[] */
    int32_t *owner;        /* TU index per 4x4 block, -1 = none                     */
    uint32_t *win_run;     /* run that claimed a window, ~0 = none                  */
    uint32_t *start;       /* 1 where a run starts; after the scan: runs before me  */
    uint32_t *runid;       /* inclusive scan of start, minus one                    */
    uint32_t *wcount;      /* wait entries per TU; after the scan: first entry      */
    uint32_t *wbegin;
    uint8_t *flags;        /* bit 0 signal, bit 1 tile_ok                           */
    uint32_t *gstart;      /* TU index where run r starts; [n_runs] = n             */
    uint32_t *wait_idx;
    u32x4 *sched, *groups;
    uint32_t *result;      /* [0] fail, [1] number of runs, [2] wait entries, [3] no wavefront keys, [4] the widest wavefront:
                              the largest number of runs that share a dependency depth */
    uint32_t wait_cap;     /* words reserved for wait_idx                           */
    uint32_t *cell_claim;  /* per 64x64-luma cell and plane: TU that opened it, ~0 = none (is the CTB 64?) */
    uint32_t *cell_edges;  /* bit 0 left, 1 above, 2 above-left, 3 above-right: cells this cell's TUs read */
    uint32_t *cell_depth;  /* longest chain of such edges ending here: the wavefront index of the cell     */
    uint32_t n_cells, cgh[3];
    uint32_t cell_off[3], cgw[3];
    int cshift[3];         /* log2 of the cell size in samples of the plane          */
    uint32_t *keys32_in, *keys32_out;      /* wavefront key (the depth of its cell) per run */
    uint32_t *vals_in, *rank_of;           /* sort payload (run); ticket of a run    */
};

struct PlanInit {
    uint32_t *p[4];
    size_t words[4];
    uint32_t value[4];
};
__global__ __launch_bounds__(256) void k_plan_init(PlanInit in) /* blockIdx.y: the region */
{
    uint32_t *p = in.p[blockIdx.y];
    const size_t words = in.words[blockIdx.y];
    const uint32_t v = in.value[blockIdx.y];
    const size_t head = (size_t)((4 - (((uintptr_t)p >> 2) & 3)) & 3); /* words in front of the first 16-byte boundary */
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    if (tid < head && tid < words) p[tid] = v;
    if (words <= head) return;
    u32x4 *q = (u32x4 *)(p + head);
    const size_t quads = (words - head) / 4;
    const u32x4 v4 = {v, v, v, v};
    for (size_t i = tid; i < quads; i += nth) q[i] = v4;
    const size_t tail = head + quads * 4;
    if (tid < words - tail) p[tail + tid] = v;
}

__device__ __forceinline__ uint32_t win_of(const PlanArgs &a, const ffhip_hevc_tu &t)
{
    const int c = t.cidx;
    return a.win_off[c] + (uint32_t)(t.y >> a.wl[c]) * (uint32_t)a.gw[c] + (uint32_t)(t.x >> a.wl[c]);
}

__global__ __launch_bounds__(256) void k_plan_owner(PlanArgs a)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const ffhip_hevc_tu t = a.tus[i];
    const int c = t.cidx, nb = (1 << t.log2_size) >> 2;
    int32_t *o = a.owner + a.owner_off[c] + (size_t)(t.y >> 2) * a.bw[c] + (t.x >> 2);
    for (int by = 0; by < nb; by++)
        for (int bx = 0; bx < nb; bx++) o[(size_t)by * a.bw[c] + bx] = (int32_t)i;
    const ffhip_hevc_tu tp = a.tus[i ? i - 1 : 0];
    a.start[i] = (i == 0 || win_of(a, t) != win_of(a, tp)) ? 1u : 0u;
    /* does the list visit every 64x64 (luma) cell in ONE stretch per plane?  Then the coding tree block is 64 and the
     * classic wavefront order over cells -- x + 2y -- is a valid ticket order (checked edge by edge in k_plan_emit) */
    const uint32_t cell = a.cell_off[c] + (uint32_t)(t.y >> a.cshift[c]) * a.cgw[c] + (uint32_t)(t.x >> a.cshift[c]);
    const uint32_t cellp = a.cell_off[tp.cidx] + (uint32_t)(tp.y >> a.cshift[tp.cidx]) * a.cgw[tp.cidx] + (uint32_t)(tp.x >> a.cshift[tp.cidx]);
    if (i == 0 || cell != cellp)
        if (atomicCAS(a.cell_claim + cell, ~0u, i) != ~0u) a.result[3] = 1; /* a cell entered twice: no wavefront keys */
}

/* longest dependency chain per cell (the wavefront index of the cell), inside ONE workgroup per plane.  Every edge a
 * cell can have points to a cell with a smaller x + 2y (left: -1, above: -2, above-left: -3, above-right: -1), so the
 * cells of one anti-diagonal x + 2y = K depend on the three diagonals before it only: ONE sweep over K, at most one cell
 * per row and step, and all a step reads of earlier results is a ring of four diagonals in LDS (ring[K & 3][row]) --
 * results go to memory and are never read back.  (The first form relaxed ALL cells until nothing changed: 0.95 ms for an
 * 8K picture.  The second kept the whole plane's depths in LDS when they fitted, 16 384 cells, and swept memory with a
 * device-scope fence per diagonal when they did not: 0.12 ms for one 8K picture, but 3.0 ms for eight pictures' worth of
 * tiles in one plane set, a third of that call.)  A lane per row of cells: planes of up to 64 rows are swept by ONE wave
 * and no barrier (LDS serves a wave in program order); taller ones by more waves and a barrier per diagonal.
 * The cells' edge bits are packed two to a byte in LDS up front (114 688 cells: 7 300 x 4 000 coding tree blocks' worth
 * would be a 450-megapixel plane); beyond that they are read from memory where they are needed.  A grid of tiles still
 * gets depth 0 at every tile's first cell, which is the point of computing depths instead of using x + 2y itself. */
#define CELLS_LDS 114688
#define DEPTH_ROWS 1024 /* rows of cells one lane set covers per pass */
template <bool FAST> /* FAST: every plane's edge bits fit the LDS form and no plane is taller than one band -- then the sweep has no LOAD from
                        memory in it, and nothing makes a step wait for the result store of the step before (one counter serves loads and
                        stores: with the slow paths' loads in the same loop every diagonal waited ~0.5 us for its own store) */
__global__ __launch_bounds__(1024) void k_plan_cell_depth(PlanArgs a)
{
    __shared__ unsigned char el[CELLS_LDS / 2];
    __shared__ unsigned short ring[4][DEPTH_ROWS];
    const int c = blockIdx.x;
    const uint32_t gw = a.cgw[c], gh = a.cgh[c], cnt = gw * gh;
    if (cnt == 0) return;
    const bool lds = FAST || cnt <= CELLS_LDS;
    uint32_t *dg = a.cell_depth + a.cell_off[c];
    const uint32_t *eg = a.cell_edges + a.cell_off[c];
    const bool one_wave = blockDim.x == 64;
    if (lds) {
#pragma unroll 4
        for (uint32_t k = threadIdx.x; 2 * k < cnt; k += blockDim.x)
            el[k] = (unsigned char)((eg[2 * k] & 15u) | (2 * k + 1 < cnt ? (eg[2 * k + 1] & 15u) << 4 : 0u));
    }
    for (uint32_t y = threadIdx.x; y < 4 * DEPTH_ROWS; y += blockDim.x) (&ring[0][0])[y] = 0;
    __syncthreads();
    if (FAST) { /* a lane per row, a step without a branch: ~40 instructions (what a lone wave pays per instruction and taken branch made the
                   general loop below 0.47 us a step: 0.36 ms for the 768 diagonals of an eight-picture grid) */
        const uint32_t yl = threadIdx.x, ym = yl ? yl - 1 : 0, rowbase = yl * gw;
        const bool row = yl < gh, hu = yl > 0;
        for (uint32_t K = 0; K <= (gw - 1) + 2 * (gh - 1); K++) {
            const uint32_t x = K - 2 * yl;          /* wraps far beyond gw where the row has not started */
            const bool valid = row && x < gw;
            const uint32_t xc = valid ? x : 0, k = rowbase + xc;
            const uint32_t e = ((uint32_t)el[k >> 1] >> (4 * (k & 1))) & 15u;
            const bool hl = xc > 0, hr = hu && xc + 1 < gw;
            const uint32_t dl_ = ring[(K - 1) & 3][yl], du_ = ring[(K - 2) & 3][ym], dul_ = ring[(K - 3) & 3][ym], dur_ = ring[(K - 1) & 3][ym];
            uint32_t v = (((e & 1u) != 0) & hl) ? dl_ + 1 : 0u;
            v = max(v, (((e & 2u) != 0) & hu) ? du_ + 1 : 0u);
            v = max(v, (((e & 4u) != 0) & hl & hu) ? dul_ + 1 : 0u);
            v = max(v, (((e & 8u) != 0) & hr) ? dur_ + 1 : 0u);
            if (valid) {
                ring[K & 3][yl] = (unsigned short)v;
                dg[k] = v;
            }
            if (one_wave) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        return;
    }
    for (uint32_t y0 = 0; y0 < gh; y0 += DEPTH_ROWS) { /* bands of DEPTH_ROWS rows of cells (one, short of a 65 536-line plane): a band reads the band above through memory */
        const uint32_t rows = gh - y0 < DEPTH_ROWS ? gh - y0 : DEPTH_ROWS;
        for (uint32_t K = 0; K <= (gw - 1) + 2 * (rows - 1); K++) {
            for (uint32_t yl = threadIdx.x; yl < rows && 2 * yl <= K; yl += blockDim.x) {
                const uint32_t x = K - 2 * yl, y = y0 + yl;
                if (x >= gw) continue;
                const uint32_t k = y * gw + x;
                const uint32_t e = lds ? ((uint32_t)el[k >> 1] >> (4 * (k & 1))) & 15u : eg[k];
                const bool hl = x > 0, hu = y > 0, hr = y > 0 && x + 1 < gw;
                uint32_t v = 0;
                if (yl > 0) { /* the row above is in the ring: diagonals K - 2 (above), K - 3 (above-left), K - 1 (above-right) */
                    const uint32_t du_ = ring[(K - 2) & 3][yl - 1], dul_ = ring[(K - 3) & 3][yl - 1], dur_ = ring[(K - 1) & 3][yl - 1];
                    if ((e & 2u) && hu) v = max(v, du_ + 1);
                    if ((e & 4u) && hl && hu) v = max(v, dul_ + 1);
                    if ((e & 8u) && hr) v = max(v, dur_ + 1);
                } else if (!FAST && hu) { /* first row of a later band: the band above finished before this one began */
                    if (e & 2u) v = max(v, dg[k - gw] + 1);
                    if ((e & 4u) && hl) v = max(v, dg[k - gw - 1] + 1);
                    if ((e & 8u) && hr) v = max(v, dg[k - gw + 1] + 1);
                }
                if ((e & 1u) && hl) v = max(v, (uint32_t)ring[(K - 1) & 3][yl] + 1);
                ring[K & 3][yl] = (unsigned short)v;
                dg[k] = v;
            }
            if (one_wave) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
                /* the ring lives in LDS: wait for LDS only.  __syncthreads() also drains the vector-memory counter, i.e. every diagonal
                 * waited for its own result store to reach memory (0.5 us a step where the step's work is 0.15) */
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        if (y0 + DEPTH_ROWS < gh) { /* the next band reads this band's last row from memory */
            __threadfence();
            __syncthreads();
        }
    }
}

/* per run: its wavefront key, the depth of its cell.  The sort behind it is stable, so runs of one cell (equal keys) keep
 * their decode order.  Only the first `m` entries exist: a plan is refused unless every window has ONE run, so there are
 * never more runs than windows -- the sort's size, known on the host (an 8K picture: 24 480 against 215 472 TUs). */
__global__ __launch_bounds__(256) void k_plan_keys(PlanArgs a, uint32_t m, uint32_t no_run)
{
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    uint32_t key = no_run; /* beyond the last run: above every depth, sorts to the end */
    if (r < a.result[1]) {
        const ffhip_hevc_tu t = a.tus[a.gstart[r]];
        const uint32_t cx = (uint32_t)(t.x >> a.cshift[t.cidx]), cy = (uint32_t)(t.y >> a.cshift[t.cidx]);
        const uint32_t depth = a.result[3] ? 0u : a.cell_depth[a.cell_off[t.cidx] + cy * a.cgw[t.cidx] + cx];
        key = depth;
    }
    a.keys32_in[r] = key;
    a.vals_in[r] = r;
}

__global__ __launch_bounds__(256) void k_plan_rank(PlanArgs a, const uint32_t *sorted_runs, uint32_t m)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (a.result[1] > m) { /* more runs than windows: some window has two (k_plan_count has refused the list already); nothing was sorted for them */
        if (k == 0) a.result[0] = 1;
        return;
    }
    const uint32_t runs = a.result[1];
    if (k >= runs) return;
    a.rank_of[sorted_runs[k]] = k;
    /* How many runs can be at work at once -- the widest wavefront, i.e. the longest stretch of equal keys in the sorted order:
     * the grouped kernel keeps only about that many of its waves (the others would hold tickets far from their turn and poll).
     * The first run of a stretch finds the stretch's end by bisection; without wavefront keys (decode order) nothing is known
     * and result[4] stays 0.  (Counted by atomics while the keys were made -- 196 k adds on the two dozen words a grid of
     * tiles has depths for -- this was 1.1 ms of an 1.8-million-TU plan, returning or not.) */
    if (a.result[3]) return;
    const uint32_t key = a.keys32_out[k];
    if (k > 0 && a.keys32_out[k - 1] == key) return;
    uint32_t lo = k, hi = runs; /* keys[lo] == key, keys[hi] != key (or hi == runs) */
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (a.keys32_out[mid] == key) lo = mid;
        else hi = mid;
    }
    atomicMax(a.result + 4, hi - k);
}

/* the TUs of other runs TU i reads; returns their number (<= 66), fills deps when not NULL */
__device__ __forceinline__ int gather_deps(const PlanArgs &a, uint32_t i, const ffhip_hevc_tu &t, uint32_t *deps, bool *tile_ok)
{
    const int c = t.cidx, n = 1 << t.log2_size, wl = a.wl[c];
    const uint32_t run = a.runid[i];
    const int wx0 = (t.x >> wl) << wl, wy0 = (t.y >> wl) << wl, wsz = 1 << wl;
    const int32_t *own = a.owner + a.owner_off[c];
    int nd = 0;
    bool ok = true;
    uint32_t last = ~0u;
    auto dep = [&](int px, int py) {
        int32_t j = own[(size_t)(py >> 2) * a.bw[c] + (px >> 2)];
        if (j >= (int32_t)i) j = -1;
        const bool mine = j >= 0 && a.runid[j] == run;
        if (j >= 0 && !mine && (uint32_t)j != last) {
            bool dup = false;
            if (deps)
                for (int q = 0; q < nd && !dup; q++) dup = deps[q] == (uint32_t)j;
            if (!dup) {
                if (deps && nd < 66) deps[nd] = (uint32_t)j;
                nd++;
                last = (uint32_t)j;
            }
        }
        if (!mine && px >= wx0 && px < wx0 + wsz && py >= wy0 && py < wy0 + wsz) ok = false;
    };
    if (t.flags & 1) dep(t.x - 1, t.y - 1);
    for (int k = 0; k < 2 * n; k += 4) {
        if ((t.avail_top >> k) & 0xf) dep(t.x + k, t.y - 1);
        if ((t.avail_left >> k) & 0xf) dep(t.x - 1, t.y + k);
    }
    if (tile_ok) *tile_ok = ok;
    return nd;
}

__global__ __launch_bounds__(256) void k_plan_count(PlanArgs a)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const ffhip_hevc_tu t = a.tus[i];
    uint32_t deps[66];
    bool ok;
    const int nd = gather_deps(a, i, t, deps, &ok); /* exact count needs the de-duplication, hence the array */
    a.wcount[i] = (uint32_t)nd;
    if (nd > 64) a.result[0] = 1; /* more than the kernel's 64 pollers: leave it to the host planner */
    atomicOr((unsigned *)(a.flags + (i & ~3u)), (ok ? 2u : 0u) << (8 * (i & 3)));
    const int c = t.cidx;
    const int cx = t.x >> a.cshift[c], cy = t.y >> a.cshift[c];
    unsigned edges = 0;
    for (int q = 0; q < nd && q < 66; q++) {
        atomicOr((unsigned *)(a.flags + (deps[q] & ~3u)), 1u << (8 * (deps[q] & 3)));
        const ffhip_hevc_tu tj = a.tus[deps[q]];
        const int dx = (tj.x >> a.cshift[c]) - cx, dy = (tj.y >> a.cshift[c]) - cy;
        if (dx == 0 && dy == 0) continue;
        if (dx == -1 && dy == 0) edges |= 1u;
        else if (dx == 0 && dy == -1) edges |= 2u;
        else if (dx == -1 && dy == -1) edges |= 4u;
        else if (dx == 1 && dy == -1) edges |= 8u;
        else a.result[3] = 1; /* a dependency no coding-tree wavefront has: keep decode order */
    }
    if (edges) atomicOr(a.cell_edges + a.cell_off[c] + (uint32_t)cy * a.cgw[c] + (uint32_t)cx, edges);
    const bool starts = i == 0 || a.runid[i] != a.runid[i - 1];
    if (starts) {
        a.gstart[a.runid[i]] = i;
        if (atomicCAS(a.win_run + win_of(a, t), ~0u, a.runid[i]) != ~0u) a.result[0] = 1; /* a window with two runs */
    }
    if (i == a.n - 1) {
        a.gstart[a.runid[i] + 1] = a.n;
        a.result[1] = a.runid[i] + 1;
    }
}

__global__ __launch_bounds__(256) void k_plan_emit(PlanArgs a, uint32_t m)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    if (a.result[1] > m) return; /* refused (k_plan_rank): the runs have no tickets, and nobody will read a schedule */
    const ffhip_hevc_tu t = a.tus[i];
    const uint32_t wb = a.wbegin[i], wc = a.wcount[i];
    if (wc) {
        uint32_t deps[66];
        gather_deps(a, i, t, deps, nullptr);
        const uint32_t my_ticket = a.rank_of[a.runid[i]];
        for (uint32_t q = 0; q < wc && q < 66; q++) {
            if (wb + q < a.wait_cap) a.wait_idx[wb + q] = deps[q]; /* beyond the reservation: the caller sees result[2] and falls back */
            if (a.rank_of[a.runid[deps[q]]] >= my_ticket) a.result[0] = 1; /* would wait for a later ticket: not with this order */
        }
    }
    const u32x4 *src = (const u32x4 *)(a.tus + i);
    const uint32_t f = a.flags[i];
    u32x4 q2;
    q2.x = wb;
    q2.y = wc | ((f & 1u) << 8) | (((f >> 1) & 1u) << 9);
    q2.z = i;
    q2.w = (a.owner_off[t.cidx] + (uint32_t)(t.y >> 2) * (uint32_t)a.bw[t.cidx] + (uint32_t)(t.x >> 2)) * 20u; /* the TU's stripe of the substitution
                                                                                                                 table (JT_STRIDE bytes per 4x4 block) */
    a.sched[(size_t)i * 3] = src[0];
    a.sched[(size_t)i * 3 + 1] = src[1];
    a.sched[(size_t)i * 3 + 2] = q2;
    if (i == a.n - 1) a.result[2] = wb + wc;
    const bool starts = i == 0 || a.runid[i] != a.runid[i - 1];
    if (starts) {
        const uint32_t r = a.runid[i];
        u32x4 g;
        g.x = i;
        g.y = a.gstart[r + 1] - i;
        g.z = (uint32_t)a.wl[t.cidx];
        g.w = 0;
        a.groups[a.rank_of[r]] = g;
    }
}

__global__ __launch_bounds__(256) void k_plan_runid(uint32_t *runid, const uint32_t *start_excl, const uint32_t *start_flag_src, uint32_t n)
{
    /* inclusive - 1 = exclusive + flag - 1 */
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) runid[i] = start_excl[i] + start_flag_src[i] - 1u;
}

/* Layout of the device scratch the caller provides (32-bit words).  sched / groups / wait_idx sit where the grouped
 * kernel expects to be told they are; everything else is planner-private. */
extern "C" size_t ffhip_hevc_plan_gpu_words(long long n_tus, const int pw[3], const int ph[3], const int wl[3])
{
    size_t blocks = 0, wins = 0;
    for (int c = 0; c < 3; c++) {
        if (pw[c] <= 0) continue;
        blocks += (size_t)((pw[c] + 3) / 4) * (size_t)((ph[c] + 3) / 4);
        wins += (size_t)(((pw[c] - 1) >> wl[c]) + 1) * (size_t)(((ph[c] - 1) >> wl[c]) + 1);
    }
    const size_t n = (size_t)n_tus;
    size_t scan_tmp = 0, sort_tmp = 0, cells = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (int)n);
    if (sort_tmp > scan_tmp) scan_tmp = sort_tmp;
    for (int c = 0; c < 3; c++)
        if (pw[c] > 0) cells += (size_t)(((pw[c] - 1) >> 4) + 1) * (size_t)(((ph[c] - 1) >> 4) + 1); /* generous: cells of >= 16 samples */
    /* sched 12n | groups 4(n+1) | wait 66... bounded by 33n in theory: sized by 8n + the fallback check | owner | win | start | startx | runid |
     * wcount | wbegin | flags | gstart | result | scan temp */
    return 12 * n + 4 * (n + 1) + 8 * n + blocks + wins + 6 * n + (n + 3) / 4 + 4 + (n + 2) + 16 + (scan_tmp + 3) / 4 + 128 + 4 * cells + 1 + 3 * n + 4 * n + 8;
}

/* Returns 0 when the plan is in place (n_groups, n_wait filled), 1 when the list needs the host planner.
 * With d_result != NULL nothing is waited for: the plan is only ENQUEUED, *d_result points at the device words
 * {refused, number of groups, wait entries} the grouped kernel reads for itself (with *wait_cap, the reservation the
 * wait entries must fit), *n_groups is left alone and the return value is 0. */
extern "C" int ffhip_hevc_plan_gpu(const ffhip_hevc_tu *d_tus, long long n_tus, const int pw[3], const int ph[3], const int wl[3],
                                   uint32_t *scratch, hipStream_t st, const u32x4 **sched, const u32x4 **groups, const uint32_t **wait_idx,
                                   int *n_groups, const uint32_t **d_result, uint32_t *wait_cap_out)
{
    PlanArgs a;
    const size_t n = (size_t)n_tus;
    a.tus = d_tus;
    a.n = (uint32_t)n;
    size_t blocks = 0, wins = 0;
    for (int c = 0; c < 3; c++) {
        a.pw[c] = pw[c]; a.ph[c] = ph[c]; a.wl[c] = wl[c];
        a.bw[c] = (pw[c] + 3) / 4;
        a.gw[c] = pw[c] > 0 ? ((pw[c] - 1) >> wl[c]) + 1 : 0;
        a.owner_off[c] = (uint32_t)blocks;
        a.win_off[c] = (uint32_t)wins;
        if (pw[c] > 0) {
            blocks += (size_t)a.bw[c] * (size_t)((ph[c] + 3) / 4);
            wins += (size_t)a.gw[c] * (size_t)(((ph[c] - 1) >> wl[c]) + 1);
        }
    }
    size_t scan_tmp = 0, sort_tmp = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (int)n);
    if (sort_tmp > scan_tmp) scan_tmp = sort_tmp;
    uint32_t *p = scratch;
    a.sched = (u32x4 *)p; p += 12 * n;
    a.groups = (u32x4 *)p; p += 4 * (n + 1);
    const size_t wait_cap = 8 * n;
    a.wait_cap = (uint32_t)wait_cap;
    a.wait_idx = p; p += wait_cap;
    a.owner = (int32_t *)p; p += blocks;
    a.win_run = p; p += wins;
    a.start = p; p += n;
    uint32_t *start_excl = p; p += n;
    a.runid = p; p += n;
    a.wcount = p; p += n;
    a.wbegin = p; p += n;
    a.gstart = p; p += n + 2;
    a.flags = (uint8_t *)p; p += (n + 3) / 4 + 4;
    a.result = p; p += 16;
    size_t cells = 0;
    for (int c = 0; c < 3; c++) {
        a.cshift[c] = c == 0 ? 6 : 6 - ((pw[c] > 0 && pw[c] * 2 <= pw[0] + 1) ? 1 : 0); /* the cell is 64x64 LUMA samples */
        a.cgw[c] = pw[c] > 0 ? (uint32_t)(((pw[c] - 1) >> a.cshift[c]) + 1) : 0;
        a.cell_off[c] = (uint32_t)cells;
        if (pw[c] > 0) cells += (size_t)a.cgw[c] * (size_t)(((ph[c] - 1) >> a.cshift[c]) + 1);
    }
    a.cell_claim = p; p += cells;
    a.cell_edges = p; p += cells;
    a.cell_depth = p; p += cells;
    p += cells + 1; /* (a histogram of runs per depth until round 3; the layout formula is shared with the host) */
    a.n_cells = (uint32_t)cells;
    for (int c = 0; c < 3; c++) a.cgh[c] = pw[c] > 0 ? (uint32_t)(((ph[c] - 1) >> a.cshift[c]) + 1) : 0;
    a.vals_in = p; p += n;
    uint32_t *vals_out = p; p += n;
    a.rank_of = p; p += n;
    p = (uint32_t *)(((uintptr_t)p + 7) & ~(uintptr_t)7);
    a.keys32_in = p; p += 2 * n;      /* (sized as in round 2, when the keys were 64-bit: the layout formula is shared with the host) */
    a.keys32_out = p; p += 2 * n;
    void *tmp = (void *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    /* owner = -1, win_run = ~0 (adjacent); flags, result = 0 (adjacent); cell_claim = ~0; cell_edges = 0 (the edge bits are OR-ed
     * in; every cell's depth is written by the sweep): ONE launch for the four regions -- as four memsets they were four more
     * kernel boundaries in front of a chain of a dozen small kernels */
    {
        PlanInit in;
        in.p[0] = (uint32_t *)a.owner; in.words[0] = blocks + wins; in.value[0] = ~0u;
        in.p[1] = (uint32_t *)a.flags; in.words[1] = (n + 3) / 4 + 4 + 16; in.value[1] = 0u;
        in.p[2] = a.cell_claim; in.words[2] = cells; in.value[2] = ~0u;
        in.p[3] = a.cell_edges; in.words[3] = cells; in.value[3] = 0u;
        size_t most = 0;
        for (int r = 0; r < 4; r++) most = in.words[r] > most ? in.words[r] : most;
        const size_t wg = (most / 4 + 255) / 256 + 1;
        hipLaunchKernelGGL(k_plan_init, dim3((unsigned)(wg > 4096 ? 4096 : wg), 4), dim3(256), 0, st, in);
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_plan_owner, dim3(grid), dim3(256), 0, st, a);
    if (hipcub::DeviceScan::ExclusiveSum(tmp, scan_tmp, a.start, start_excl, (int)n, st) != hipSuccess) return FFHIP_EIO;
    hipLaunchKernelGGL(k_plan_runid, dim3(grid), dim3(256), 0, st, a.runid, start_excl, a.start, (uint32_t)n);
    hipLaunchKernelGGL(k_plan_count, dim3(grid), dim3(256), 0, st, a);
    if (hipcub::DeviceScan::ExclusiveSum(tmp, scan_tmp, a.wcount, a.wbegin, (int)n, st) != hipSuccess) return FFHIP_EIO;
    /* tickets: runs sorted by (wavefront key of their cell, decode order) */
    {   /* one diagonal per step, at most one cell per row of cells, a lane per row: a picture of up to 64 rows of cells is swept by ONE
         * wave per plane (a wave-local barrier per step), taller ones by as many waves as they have rows (up to 1024 threads) and a
         * workgroup barrier per step.  (One wave with three rows per lane: 1.15 us a step on the 144-row plane of an eight-picture grid.) */
        uint32_t max_gh = 0;
        for (int c = 0; c < 3; c++) max_gh = a.cgh[c] > max_gh ? a.cgh[c] : max_gh;
        const unsigned threads = max_gh >= 1024 ? 1024u : (unsigned)((max_gh + 63) / 64 * 64);
        bool fast = max_gh <= DEPTH_ROWS;
        for (int c = 0; c < 3; c++) fast = fast && (size_t)a.cgw[c] * a.cgh[c] <= CELLS_LDS;
        if (fast) hipLaunchKernelGGL(k_plan_cell_depth<true>, dim3(3), dim3(threads ? threads : 64u), 0, st, a);
        else hipLaunchKernelGGL(k_plan_cell_depth<false>, dim3(3), dim3(threads ? threads : 64u), 0, st, a);
    }
    const size_t m = n < wins ? n : wins; /* runs <= windows, or the plan is refused (k_plan_count: a window with two runs) */
    int key_bits = 1;
    while (key_bits < 32 && (1ull << key_bits) <= cells + 1) key_bits++; /* depths are < cells; 2^key_bits - 1 stands for "no run" and sorts last */
    hipLaunchKernelGGL(k_plan_keys, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, (uint32_t)m, (uint32_t)((1ull << key_bits) - 1));
    if (hipcub::DeviceRadixSort::SortPairs(tmp, scan_tmp, a.keys32_in, a.keys32_out, a.vals_in, vals_out, (int)m, 0, key_bits, st) != hipSuccess) return FFHIP_EIO;
    hipLaunchKernelGGL(k_plan_rank, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, (const uint32_t *)vals_out, (uint32_t)m);
    hipLaunchKernelGGL(k_plan_emit, dim3(grid), dim3(256), 0, st, a, (uint32_t)m);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    if (d_result) {
        *sched = a.sched;
        *groups = a.groups;
        *wait_idx = a.wait_idx;
        *d_result = a.result;
        if (wait_cap_out) *wait_cap_out = (uint32_t)wait_cap;
        return 0;
    }
    uint32_t res[3] = {1, 0, 0};
    FFHIP_CHECK(hipMemcpyAsync(res, a.result, sizeof res, hipMemcpyDeviceToHost, st), FFHIP_EIO);
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
    if (res[0] || res[2] > wait_cap) return 1; /* not contiguous, too many pollers, or more wait entries than reserved */
    *sched = a.sched;
    *groups = a.groups;
    *wait_idx = a.wait_idx;
    *n_groups = (int)res[1];
    return 0;
}
