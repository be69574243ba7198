/*
 * ffhip_hevc_plan_gpu.hip -- the group schedule of ffhip_hevc_intra_recon built ON the device.
 *
 * The host planner (plan_groups in ffhip_hevc_intra.hip) costs ~30 ns per TU on one core: more than the kernel it
 * feeds once a picture has a few hundred thousand TUs (an 8K picture, a HEIF grid).  Everything it derives is a pure
 * function of the TU list, one TU at a time, given the map "which TU owns this 4x4 block":
 *   k_plan_owner   every TU stamps its index on its blocks, says whether it starts a new RUN (a maximal stretch of consecutive TUs whose
 *                  top-left corners fall into the same window of the same plane) -- per block of 256 TUs the number of run starts -- and
 *                  marks which neighbouring 64x64 cells its available neighbours lie in (the cells' dependency edges)
 *   k_plan_scan    exclusive scan of those block totals (256 entries x 16 per workgroup, no chain between workgroups)
 *   k_plan_runid   run id per TU: the block's prefix + the starts in front of it inside the block (ballots)
 *   k_plan_count   per TU, ONCE: the TUs of OTHER runs among its available neighbours (only earlier ones count: a block
 *                  stamped by a later TU held older content when the sequential decoder looked) -- written straight into
 *                  the wait list, whose room a block reserves with one atomic add (where a TU's entries sit does not
 *                  matter, only that they are together); whether all its in-window neighbours are its own run's (LDS tile
 *                  allowed), who must publish a done flag; per run start: its position, the claim on its window -- a window
 *                  claimed by two runs means the list is not "groups contiguous in decode order" -- and the count of runs
 *                  per 64x64 cell
 *   k_plan_cell_depth_rows / k_plan_cell_depth   the wavefront index of every cell (longest chain of dependency edges ending there):
 *                  row by row, a prefix scan per row of cells; for planes wider than 1024 cells one anti-diagonal per step
 *   tickets        runs in (depth of their cell, decode order) order WITHOUT a sort: a cell is entered once, so its runs are
 *                  consecutive run ids and stay in decode order inside a contiguous range of tickets; the cells of one depth
 *                  never depend on each other, so their ranges may follow each other in any order -- a histogram of runs per
 *                  depth (k_plan_cell_hist), its exclusive scan (k_plan_scan, which also finds the widest wavefront), one
 *                  atomic add per cell for the cell's range (k_plan_cell_base), and ticket = range start + run - first run of
 *                  the cell (k_plan_rank).  Until round 4 this was hipcub's radix sort over (depth, run) pairs and two hipcub
 *                  scans over all TUs: a dozen rocprim launches per picture.
 *   k_plan_emit    schedule slots (position = TU index), group records, and the check that every TU a TU waits for has a
 *                  smaller ticket -- the deadlock-freedom condition of the grouped kernel -- from the wait list as written
 * Same output layout as the host planner.  Nothing here is a library primitive: the scans are a wave shuffle scan + LDS.
 */
#include "ffhip_internal.h"

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
                              the largest number of runs that share a dependency depth, [5] log2 of the luma window, [6] a record failed
                              k_hevc_check_tus, [7] the list was sorted by plane (k_part_*) */
    uint32_t wait_cap;     /* words reserved for wait_idx                           */
    uint32_t wsub_n;       /* slices in use: a power of two, at most PLAN_WSUB, never more than blocks of 256 TUs */
    uint32_t *wsub;        /* PLAN_WSUB counters, one per 128-byte line: wait entries handed out of slice r of wait_idx */
    uint32_t *cell_claim;  /* per 64x64-luma cell and plane: TU that opened it, ~0 = none (is the CTB 64?) */
    uint32_t *cell_edges;  /* bit 0 left, 1 above, 2 above-left, 3 above-right: cells this cell's TUs read */
    uint32_t *cell_depth;  /* longest chain of such edges ending here: the wavefront index of the cell     */
    uint32_t n_cells, cgh[3];
    uint32_t cell_off[3], cgw[3];
    int cshift[3];         /* log2 of the cell size in samples of the plane          */
    uint32_t *rank_of;     /* ticket of a run                                          */
    uint32_t *blk_tot;     /* per block of 256 TUs: run starts                                        */
    uint32_t *blk_pre;     /* its exclusive scan: starts in the blocks before                         */
    uint32_t *cell_nruns;  /* runs per cell                                            */
    uint32_t *cell_base;   /* first ticket of the cell's runs                          */
    uint32_t *hist;        /* [depths][shards] runs per (depth, shard)                 */
    uint32_t *hist_pre;    /* its exclusive scan: first ticket of the pair             */
    uint32_t *fill;        /* [depths][shards] tickets of the pair handed out so far   */
    uint32_t depths;       /* a bound on the depths: a chain ending at cell (x, y) has at most x + 2y edges */
    float stripe_scale[3]; /* 2^shard_log2 / cells of the plane */
    uint32_t shard_log2;   /* the counters of one depth are spread over 2^shard_log2 words, picked by the cell's block: a grid of tiles has
                              two dozen distinct depths for its 200 000 cells, and that many atomic adds on two dozen words took 0.4 ms */
    /* the list sorted by plane (k_part_*): the caller's records, the copy the planner and everything behind it work on, and per
     * (plane, block of 256 records) the records of that plane in the block / in front of it in the sorted list */
    const ffhip_hevc_tu *raw;
    ffhip_hevc_tu *sorted;
    uint32_t *part_tot, *part_pre;
    uint32_t part_nb;
};

/* ---- scans: a shuffle scan inside the wave, the waves' totals through LDS ---- */
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, const int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
/* exclusive scan of one value per thread over the workgroup (blockDim.x a multiple of 64, at most 1024); *total = the workgroup's sum.
 * `wsum` is LDS of 17 words; two barriers */
__device__ __forceinline__ uint32_t block_excl_scan(const uint32_t v, uint32_t *wsum, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (w == 0) {
        const uint32_t x = lane < nw ? wsum[lane] : 0u;
        const uint32_t xi = wave_incl_scan(x, lane);
        if (lane < nw) wsum[lane] = xi - x;
        if (lane == nw - 1) wsum[16] = xi;
    }
    __syncthreads();
    const uint32_t r = wsum[w] + inc - v;
    *total = wsum[16];
    return r;
}
/* Exclusive scan of v[0..n) into out[0..n) (out != v), 4096 entries per workgroup.  A workgroup first adds up everything IN FRONT of its
 * chunk by itself -- the tables scanned here are small (n / 256 block totals, at most 32 K histogram words), reading them again costs a
 * few microseconds and no workgroup waits for another -- then scans its chunk.  (ONE workgroup walking the table pass by pass, until late in
 * round 4: 58 us for the 7 176 block totals and 148 us for the histogram of an eight-picture grid, the side stream's kernels on the same
 * CUs.)  With vmax: v is [groups][1 << group_log2] and *vmax (zero beforehand) receives the largest GROUP total (the widest wavefront:
 * the most runs of one depth). */
__global__ __launch_bounds__(256) void k_plan_scan(const uint32_t *v, uint32_t *out, uint32_t n, uint32_t *vmax, uint32_t group_log2, const uint32_t *skip)
{
    /* 256 threads, sixteen entries each: a workgroup of 1024 needs sixteen free wave slots on ONE CU at once, and next to the side stream's
     * thousands of four-wave workgroups it waited for them (the same scan: 5 us alone, 60 - 130 us there) */
    __shared__ uint32_t wsum[17];
    __shared__ uint32_t red[4];
    if (skip && *skip) return; /* no wavefront keys: nobody reads the histogram */
    __builtin_amdgcn_s_setprio(3);
    const uint32_t base = blockIdx.x * 4096u;
    uint32_t s = 0;
    { /* eight loads in flight per thread: one after the other they were up to 112 trips to memory in a row (0.2 ms next to a kernel that fills the chip) */
        uint32_t s8[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        for (uint32_t i = threadIdx.x; i < base; i += 8 * 256) {
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) s8[u] += i + u * 256 < base ? v[i + u * 256] : 0u;
        }
        s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const uint32_t carry = red[0] + red[1] + red[2] + red[3];
    const uint32_t i = base + 16 * threadIdx.x;
    uint32_t x[16], mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { x[k] = i + k < n ? v[i + k] : 0u; mine += x[k]; }
    uint32_t total;
    uint32_t run = carry + block_excl_scan(mine, wsum, &total);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (i + k < n) out[i + k] = run;
        run += x[k];
    }
    if (vmax) { /* group totals from what the threads hold (a loop of loads per group was up to 64 trips to memory in a row) */
        __shared__ uint32_t tsum[256];
        uint32_t mx = 0;
        if (group_log2 >= 4) { /* a group is 1, 2 or 4 threads' sixteen entries */
            const uint32_t gt = 1u << (group_log2 - 4);
            tsum[threadIdx.x] = mine;
            __syncthreads();
            if ((threadIdx.x & (gt - 1)) == 0)
                for (uint32_t q = 0; q < gt; q++) mx += tsum[threadIdx.x + q];
        } else { /* several groups inside a thread's sixteen entries */
            const uint32_t gs = 1u << group_log2;
#pragma unroll
            for (uint32_t g0 = 0; g0 < 16; g0++) {
                uint32_t t = 0;
                if ((g0 & (gs - 1)) == 0) {
#pragma unroll
                    for (uint32_t q = 0; q < 8; q++) t += (q < gs && g0 + q < 16) ? x[(g0 + q) & 15] : 0u;
                    mx = t > mx ? t : mx;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)mx, o, 64); mx = t > mx ? t : mx; }
        if ((threadIdx.x & 63) == 0 && mx) atomicMax(vmax, mx);
    }
}

struct PlanInit {
    uint32_t *p[5];
    size_t words[5];
    uint32_t value[5];
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

/* ---- the list sorted by plane ----
 * The reference decodes coding unit by coding unit -- the unit's luma tree, then its Cb tree, then its Cr tree
 * (decode_cu_coded_intra_prediction_mode, coding/hevc.c:5013-5180, calling decode_intra_block :4665-4805 once per component) -- so a coding
 * tree block with more than one coding unit visits its 64x64 area of every plane several times, with the other planes in between.  A run, a
 * window visit and a cell visit as everything below defines them ("consecutive records of the list") would end at every such switch: seven
 * of seven lists the reference's own decoder recorded had one group per 8x8 window and decode-order tickets that way.  The planes never read
 * each other in this interface (cross-component prediction as the reference calls it reads the chroma block itself), so any order that
 * keeps each plane's own subsequence decodes to the same samples: the planner, the programs and the grouped kernel work on a STABLE
 * partition of the list by plane -- all luma records in list order, then Cb, then Cr.  A counting sort with three keys: records per (plane,
 * block of 256), the exclusive scan of that table laid out plane by plane (k_plan_scan), and the scatter. */
__device__ __forceinline__ uint32_t part_plane_of(const ffhip_hevc_tu *tus, const uint32_t i)
{
    const uint32_t w1 = ((const uint32_t *)(tus + i))[1]; /* log2_size | cidx << 8 | pred_mode << 16 | flags << 24 */
    const uint32_t c = (w1 >> 8) & 0xffu;
    return c > 2u ? 2u : c; /* (a record k_hevc_check_tus refuses: the copy is never read) */
}
/* ranks inside the block: what each of the three planes has in the waves in front of mine, and in the lanes in front of me */
__device__ __forceinline__ uint32_t part_rank(const uint32_t c, const bool live, uint32_t (*wtot)[3], uint32_t tot[3])
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long b[3];
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) b[k] = __builtin_amdgcn_ballot_w64(live && c == k);
    if (lane == 0)
        for (int k = 0; k < 3; k++) wtot[w][k] = (uint32_t)__popcll(b[k]);
    __syncthreads();
    const unsigned long long mine = c == 0 ? b[0] : (c == 1 ? b[1] : b[2]);
    uint32_t r = (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
    for (int q = 0; q < 4; q++) {
        if (q < w) r += wtot[q][c];
#pragma unroll
        for (int k = 0; k < 3; k++) tot[k] = (q ? tot[k] : 0u) + wtot[q][k];
    }
    return r;
}
__global__ __launch_bounds__(256) void k_part_count(PlanArgs a)
{
    __shared__ uint32_t wtot[4][3];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.n;
    uint32_t tot[3];
    part_rank(live ? part_plane_of(a.raw, i) : 0u, live, wtot, tot);
    if (threadIdx.x < 3) a.part_tot[threadIdx.x * a.part_nb + blockIdx.x] = tot[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_part_scatter(PlanArgs a)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    __shared__ uint32_t wtot[4][3];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.n;
    const uint32_t c = live ? part_plane_of(a.raw, i) : 0u;
    uint32_t tot[3];
    const uint32_t r = part_rank(c, live, wtot, tot);
    if (!live) return;
    const uint32_t dst = a.part_pre[c * a.part_nb + blockIdx.x] + r;
    const u32x4 *src = (const u32x4 *)(a.raw + i);
    u32x4 *out = (u32x4 *)(a.sorted + dst);
    const u32x4 q0 = src[0], q1 = src[1];
    out[0] = q0; out[1] = q1;
    if (i == 0) a.result[7] = 1u; /* (diagnostics: the planner worked on the sorted copy) */
}

__device__ __forceinline__ uint32_t win_of(const PlanArgs &a, const ffhip_hevc_tu &t)
{
    const int c = t.cidx;
    return a.win_off[c] + (uint32_t)(t.y >> a.wl[c]) * (uint32_t)a.gw[c] + (uint32_t)(t.x >> a.wl[c]);
}

__device__ __forceinline__ bool plan_owner_tu(const PlanArgs &a, const uint32_t i)
{
    const ffhip_hevc_tu t = a.tus[i];
    const int c = t.cidx, nb = (1 << t.log2_size) >> 2;
    int32_t *o = a.owner + a.owner_off[c] + (size_t)(t.y >> 2) * a.bw[c] + (t.x >> 2);
    /* rows of a 16x16 / 32x32 TU are one / two 16-byte stores, of an 8x8 TU one 8-byte store, where the plane's block rows keep that alignment
     * (a 32x32 TU was 64 stores from one lane; the 25 M blocks of an eight-picture grid took 0.17 ms) */
    const uintptr_t al = (uintptr_t)o | ((uintptr_t)a.bw[c] << 2);
    if (nb >= 4 && !(al & 15)) {
        const u32x4 v4 = {i, i, i, i};
        for (int by = 0; by < nb; by++)
            for (int bx = 0; bx < nb; bx += 4) *(u32x4 *)(o + (size_t)by * a.bw[c] + bx) = v4;
    } else if (nb == 2 && !(al & 7)) {
        const u32x2 v2 = {i, i};
        *(u32x2 *)o = v2;
        *(u32x2 *)(o + a.bw[c]) = v2;
    } else {
        for (int by = 0; by < nb; by++)
            for (int bx = 0; bx < nb; bx++) o[(size_t)by * a.bw[c] + bx] = (int32_t)i;
    }
    const ffhip_hevc_tu tp = a.tus[i ? i - 1 : 0];
    const bool starts = i == 0 || win_of(a, t) != win_of(a, tp);
    a.start[i] = starts ? 1u : 0u;
    /* does the list visit every 64x64 (luma) cell in ONE stretch per plane?  Then the coding tree block is 64 and the
     * classic wavefront order over cells -- x + 2y -- is a valid ticket order (checked edge by edge in k_plan_emit) */
    const uint32_t cell = a.cell_off[c] + (uint32_t)(t.y >> a.cshift[c]) * a.cgw[c] + (uint32_t)(t.x >> a.cshift[c]);
    const uint32_t cellp = a.cell_off[tp.cidx] + (uint32_t)(tp.y >> a.cshift[tp.cidx]) * a.cgw[tp.cidx] + (uint32_t)(tp.x >> a.cshift[tp.cidx]);
    if (i == 0 || cell != cellp)
        if (atomicCAS(a.cell_claim + cell, ~0u, i) != ~0u) a.result[3] = 1; /* a cell entered twice: no wavefront keys */
    /* Which neighbouring cells this cell's TUs read -- the edges the depth sweep runs over -- from where the TU's AVAILABLE neighbours lie:
     * geometry and the masks alone.  (k_plan_count found them among the dependencies it counted, earlier TUs of other runs only; this is a
     * superset -- a neighbour marked available that no earlier TU wrote adds an edge the wait list will not have, which only deepens a cell --
     * and it lets the sweep start behind THIS kernel, next to k_plan_count, instead of behind that one.)  A neighbour in a cell no
     * coding-tree wavefront has in front of this one (right of it in the same row, or below) keeps the tickets in decode order. */
    {
        const int cs = 1 << a.cshift[c], lx = t.x & (cs - 1), ly = t.y & (cs - 1), n2 = 2 << t.log2_size;
        const unsigned long long span = n2 >= 64 ? ~0ull : (1ull << n2) - 1;
        const unsigned long long top = t.avail_top & span, left = t.avail_left & span;
        const int in_x = cs - lx, in_y = cs - ly; /* neighbour columns / rows that still belong to my cell's column / row of cells */
        const unsigned long long top_in = in_x >= 64 ? top : top & ((1ull << in_x) - 1), left_in = in_y >= 64 ? left : left & ((1ull << in_y) - 1);
        unsigned edges = 0;
        bool odd = (left != left_in);                    /* left of me, but in the row of cells below */
        if (ly == 0) edges |= (top_in ? 2u : 0u) | (top != top_in ? 8u : 0u);
        else odd |= (top != top_in);                     /* the cell to the right, same row */
        if (lx == 0 && left_in) edges |= 1u;
        if (t.flags & 1) edges |= (lx == 0 && ly == 0) ? 4u : (lx == 0 ? 1u : (ly == 0 ? 2u : 0u));
        if (odd) a.result[3] = 1;
        if (edges) atomicOr(a.cell_edges + cell, edges);
    }
    return starts;
}

__global__ __launch_bounds__(256) void k_plan_owner(PlanArgs a)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    __shared__ uint32_t wtot[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    bool starts = false;
    if (i < a.n) starts = plan_owner_tu(a, i);
    /* run starts of this block of 256 TUs, for the scan behind */
    const unsigned long long b = __builtin_amdgcn_ballot_w64(starts);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) a.blk_tot[blockIdx.x] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
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
#define DEPTH_SKEW 16
#define DEPTH_RING 64
template <bool FAST> /* FAST: every plane's edge bits fit the LDS form and no plane is taller than one band -- then the sweep has no LOAD from
                        memory in it, and nothing makes a step wait for the result store of the step before (one counter serves loads and
                        stores: with the slow paths' loads in the same loop every diagonal waited ~0.5 us for its own store) */
__global__ __launch_bounds__(1024) void k_plan_cell_depth(PlanArgs a)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    __builtin_amdgcn_s_setprio(3); /* three workgroups next to the side stream's thousands of waves: first in line at the issue arbiter */
    __shared__ unsigned char el[CELLS_LDS / 2];
    __shared__ unsigned short ring[4][DEPTH_ROWS];
    unsigned short (*edge)[DEPTH_RING] = (unsigned short (*)[DEPTH_RING]) & ring[0][0]; /* the FAST sweep's use of the same LDS: [wave][diagonal mod DEPTH_RING] */
    const int c = blockIdx.x;
    const uint32_t gw = a.cgw[c], gh = a.cgh[c], cnt = gw * gh;
    if (cnt == 0) return;
    const bool lds = FAST || cnt <= CELLS_LDS;
    uint32_t *dg = a.cell_depth + a.cell_off[c];
    const uint32_t *eg = a.cell_edges + a.cell_off[c];
    const bool one_wave = blockDim.x == 64;
    if (lds) {
        /* (the FAST form is launched with 1024 threads whatever the plane's height: 192 threads fetching the 69 120 cells of an eight-picture
         * grid two at a time WERE most of the kernel -- 0.2 ms; the waves the sweep has no rows for leave behind the barrier below) */
#pragma unroll 4
        for (uint32_t k = threadIdx.x; 2 * k < cnt; k += blockDim.x) {
            uint32_t nib[2];
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) { /* bit 0 left, 1 above, 2 above-left, 3 above-right: not where the plane has no such cell */
                const uint32_t kk = 2 * k + h, cy = kk / gw, cx = kk - cy * gw;
                const uint32_t keep = (cx > 0 ? 5u : 0u) | (cy > 0 ? 2u : 0u) | ((cy > 0 && cx + 1 < gw) ? 8u : 0u);
                const uint32_t both = (cx > 0 && cy > 0) ? 4u : 0u;
                nib[h] = kk < cnt ? eg[kk] & ((keep & ~4u) | both) : 0u;
            }
            el[k] = (unsigned char)(nib[0] | (nib[1] << 4));
        }
    }
    for (uint32_t y = threadIdx.x; y < 4 * DEPTH_ROWS; y += blockDim.x) (&ring[0][0])[y] = 0;
    __syncthreads();
    if (FAST) {
        /* A lane per row, and NOTHING of the recurrence in memory: a cell's left neighbour is the lane's own value of the step before; the
         * three above it (above-right, above, above-left: diagonals K - 1, K - 2, K - 3 of the row above) are the value the lane above
         * computed in the step before -- one DPP move -- and what that move returned one and two steps ago.  Across waves the lane above is
         * lane 63 of the previous wave: wave w runs DEPTH_SKEW steps behind wave w - 1, takes that lane's values from a ring in LDS, and a
         * barrier every DEPTH_SKEW steps keeps the distance (a value is read DEPTH_SKEW steps after it was written and overwritten
         * DEPTH_RING steps after, DEPTH_RING >= 2 DEPTH_SKEW + 2).  A lone wave pays for every instruction it issues, so the step is kept
         * short: the edge bits a cell cannot have (left of column 0, above row 0, right of the last column) are cleared once, up front;
         * the bits and the ring value of the NEXT step are fetched during this one; a term is one multiply-add by its bit; the result store
         * of a lane without a cell is dropped by its buffer offset.  (A ring of four diagonals in LDS and a barrier per step, until late in
         * round 4: 0.3 us a step, 247 us for the 766 diagonals of an eight-picture grid, 72 us for one picture's 253.) */
        const uint32_t yl = threadIdx.x, w = yl >> 6, lane = yl & 63, nw = (gh + 63) >> 6;
        if (w >= nw) return; /* (a wave that has ended is not waited for at a barrier) */
        const bool row = yl < gh;
        const uint32_t rowbase = row ? yl * gw : 0u, gwv = row ? gw : 0u; /* a lane beyond the last row never has a cell */
        const uint32_t iters = ((gw - 1) + 2 * (gh - 1) + 1 + (nw - 1) * DEPTH_SKEW + 3) & ~3u; /* a multiple of four: results leave in fours */
        const __amdgpu_buffer_rsrc_t drs = ffhip_rsrc(dg, cnt * 4u);
        const unsigned short *up_edge = edge[w ? w - 1 : 0];
        unsigned short *my_edge = edge[w];
        const bool l0 = lane == 0 && w > 0, l63 = lane == 63 && nw > 1;
        auto load_e = [&](const uint32_t xx) -> uint32_t {
            const bool v = xx < gwv;
            const uint32_t k = rowbase + (v ? xx : 0u);
            const uint32_t b = (uint32_t)el[k >> 1] >> (4 * (k & 1));
            return v ? b & 15u : 0u;
        };
        uint32_t K = 0u - w * DEPTH_SKEW; /* wraps far beyond every diagonal while the wave has not started */
        uint32_t x = K - 2 * yl;          /* ... and where the row has not started */
        uint32_t h1 = 0, n2 = 0, n3 = 0, d4[4] = {0u, 0u, 0u, 0u};
        uint32_t e = load_e(x), ev = up_edge[(K - 1) & (DEPTH_RING - 1)];
        for (uint32_t T = 0; T < iters; T++) {
            const uint32_t en = load_e(x + 1), evn = up_edge[K & (DEPTH_RING - 1)];
            uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h1, 0x138, 0xf, 0xf, false); /* wave_shr:1 -- lane i - 1 */
            n1 = l0 ? ev : n1;
            const uint32_t b0 = e & 1u, b1 = (e >> 1) & 1u, b2 = (e >> 2) & 1u, b3 = e >> 3;
            const uint32_t v = max(max(__umul24(h1, b0) + b0, __umul24(n2, b1) + b1), max(__umul24(n3, b2) + b2, __umul24(n1, b3) + b3));
            /* results leave four steps at a time (a lane's cells of consecutive steps are consecutive words): a store per step was a vector-memory
             * instruction per step that had to find room in the CU's memory pipeline -- next to a kernel that keeps that pipeline full (the
             * sweep runs beside k_plan_count) the sweep took four times as long as alone */
            d4[T & 3] = v;
            if ((T & 3) == 3) {
                const uint32_t x0 = x - 3; /* the four cells x0 .. x0 + 3 (x0 may have wrapped: then none or some are cells) */
                const bool all4 = x0 < gwv && x < gwv;
                if (__builtin_amdgcn_ballot_w64(!all4 && (x0 < gwv || x < gwv || x0 + 1 < gwv || x0 + 2 < gwv)) == 0) {
                    const u32x4 q = {d4[0], d4[1], d4[2], d4[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(q, drs, all4 ? 4u * (rowbase + x0) : 0x80000000u, 0, 0);
                } else {
#pragma unroll
                    for (uint32_t e = 0; e < 4; e++)
                        __builtin_amdgcn_raw_buffer_store_b32(d4[e], drs, x0 + e < gwv ? 4u * (rowbase + x0 + e) : 0x80000000u, 0, 0);
                }
            }
            if (l63) my_edge[K & (DEPTH_RING - 1)] = (unsigned short)v;
            n3 = n2; n2 = n1; h1 = v; e = en; ev = evn;
            x++; K++;
            if (nw > 1 && ((T + 1) & (DEPTH_SKEW - 1)) == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

/* The same depths ROW BY ROW, for planes of up to 1024 columns of cells (a 65 536-sample line): a lane per column, one step per row of cells
 * -- gh steps where the diagonal sweep above takes gw + 2 gh (an eight-picture grid of 480 x 144 cells: 144 against 766; the sweep is a chain
 * of dependent instructions in three workgroups, and next to a kernel that fills the chip every instruction of it waits its turn: 130 us
 * alone, 360 - 510 us there).  Inside a row the only dependency is on the cell to the left, d[x] = max(a[x], d[x - 1] + 1) where the cell has
 * that edge, a[x] from the row above: a prefix scan over functions d -> max(A, d + B), which compose to functions of the same form.
 * An element is (A, Be): Be = 0 "no edge to the left" (the function is the constant A), else the shift + 1. */
__device__ __forceinline__ void depth_compose(uint32_t &A2, uint32_t &B2e, const uint32_t A1, const uint32_t B1e) /* (2) after (1), into (2) */
{
    const uint32_t sh2 = B2e - 1;
    const uint32_t A = max(A2, A1 + sh2), Be = B1e ? B1e + sh2 : 0u;
    A2 = B2e ? A : A2;
    B2e = B2e ? Be : 0u;
}
__global__ __launch_bounds__(1024) void k_plan_cell_depth_rows(PlanArgs a)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    __shared__ unsigned char el[CELLS_LDS / 2];
    __shared__ unsigned short rowbuf[2][1024 + 2];
    __shared__ uint32_t wtA[16], wtB[16];
    const int c = blockIdx.x;
    const uint32_t gw = a.cgw[c], gh = a.cgh[c];
    if (gw == 0 || gh == 0) return;
    const uint32_t x = threadIdx.x, w = x >> 6, lane = x & 63, nw = (gw + 63) >> 6, cnt = gw * gh;
    uint32_t *dg = a.cell_depth + a.cell_off[c];
    const uint32_t *eg = a.cell_edges + a.cell_off[c];
    for (uint32_t i = x; i < 2 * (1024 + 2); i += blockDim.x) (&rowbuf[0][0])[i] = 0;
    /* the plane's edge bits, two cells to a byte, fetched by the whole workgroup up front: a load from memory per row -- even one issued a
     * row ahead -- was a trip to memory per step of the chain (3.5 us a row next to a kernel that fills the chip) */
#pragma unroll 4
    for (uint32_t k = x; 2 * k < cnt; k += blockDim.x)
        el[k] = (unsigned char)((eg[2 * k] & 15u) | (2 * k + 1 < cnt ? (eg[2 * k + 1] & 15u) << 4 : 0u));
    __syncthreads();
    if (w >= nw) return; /* (a wave that has ended is not waited for at a barrier) */
    const bool col = x < gw;
    /* edge bits a cell cannot have, by column: left / above-left of column 0, above-right of the last column (the row above is all zeros
     * for row 0: its bits do no harm) */
    const uint32_t keep = col ? ((x > 0 ? 5u : 0u) | 2u | (x + 1 < gw ? 8u : 0u)) : 0u;
    auto bits = [&](const uint32_t y) -> uint32_t {
        const uint32_t k = (col && y < gh) ? y * gw + x : 0u;
        return ((uint32_t)el[k >> 1] >> (4 * (k & 1))) & keep;
    };
    uint32_t e = bits(0) & ~14u; /* row 0: nothing above */
    for (uint32_t y = 0; y < gh; y++) {
        const unsigned short *prev = rowbuf[y & 1];
        unsigned short *cur = rowbuf[(y & 1) ^ 1];
        const uint32_t en = bits(y + 1); /* the next row's bits */
        const uint32_t pl = prev[x], pu = prev[x + 1], pr = prev[x + 2]; /* rowbuf[.][1 + column] */
        uint32_t A = max(max((e & 2u) ? pu + 1 : 0u, (e & 4u) ? pl + 1 : 0u), (e & 8u) ? pr + 1 : 0u);
        uint32_t Be = (e & 1u) ? 2u : 0u;
        /* inclusive scan inside the wave: offsets 1 .. 8 inside rows of sixteen lanes (DPP row_shr), then the rows one after the other */
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            uint32_t A1, B1;
            if (o == 1) { A1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, 0x111, 0xf, 0xf, false); B1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Be, 0x111, 0xf, 0xf, false); }
            else if (o == 2) { A1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, 0x112, 0xf, 0xf, false); B1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Be, 0x112, 0xf, 0xf, false); }
            else if (o == 4) { A1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, 0x114, 0xf, 0xf, false); B1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Be, 0x114, 0xf, 0xf, false); }
            else { A1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, 0x118, 0xf, 0xf, false); B1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Be, 0x118, 0xf, 0xf, false); }
            uint32_t A2 = A, B2 = Be;
            depth_compose(A2, B2, A1, B1);
            const bool has = (int)(lane & 15) >= o;
            A = has ? A2 : A; Be = has ? B2 : Be;
        }
#pragma unroll
        for (int r = 1; r < 4; r++) { /* rows 1, 2, 3 of the wave take the (finished) last lane of the row before */
            const uint32_t A1 = (uint32_t)__builtin_amdgcn_readlane((int)A, 16 * r - 1), B1 = (uint32_t)__builtin_amdgcn_readlane((int)Be, 16 * r - 1);
            uint32_t A2 = A, B2 = Be;
            depth_compose(A2, B2, A1, B1);
            const bool mine = (int)(lane >> 4) == r;
            A = mine ? A2 : A; Be = mine ? B2 : Be;
        }
        if (nw > 1) { /* the waves in front of mine: their totals through LDS, composed in order */
            if (lane == 63) { wtA[w] = A; wtB[w] = Be; }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            uint32_t PA = 0, PB = 0; /* of nothing: the constant 0 */
            for (uint32_t q = 0; q < w; q++) {
                uint32_t A2 = wtA[q], B2 = wtB[q];
                depth_compose(A2, B2, PA, PB);
                PA = A2; PB = B2;
            }
            if (w) depth_compose(A, Be, PA, PB);
        }
        /* (the row's first cell has no edge to the left: every prefix is a constant by now, A is the depth) */
        if (col) {
            cur[x + 1] = (unsigned short)A;
            dg[(size_t)y * gw + x] = A;
        }
        e = en;
        if (nw > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

/* Tickets without a sort.  A cell is entered ONCE (or result[3] says otherwise and tickets stay in decode order), so the runs of a cell are
 * consecutive run ids, and all of them have the cell's depth.  What the grouped kernel needs of the ticket order is (1) every run a TU waits
 * for has a smaller ticket and (2) tickets roughly follow the ready front.  Dependencies point to cells of smaller depth or to earlier runs
 * of the same cell, never to another cell of the same depth: so the depths take their tickets in order (an exclusive scan of "runs per
 * depth"), the cells of one depth take contiguous ranges in WHATEVER order their atomic adds arrive, and the runs of a cell keep their decode
 * order inside the cell's range.  m bounds the runs: a plan is refused unless every window has ONE run, so there are never more runs than
 * windows -- known on the host (an 8K picture: 24 480 against 215 472 TUs). */
/* the counter a cell's runs are counted in: (depth, stripe) -- the stripe is the cell's position in its plane in 2^shard_log2 steps, so inside
 * a depth the tickets still run from the top of the picture to its bottom, luma and chroma of one area side by side: a group and the groups
 * it waits for (one depth up, a neighbouring cell) then sit at about the same place of their depths' ranges, a whole range of tickets apart.
 * (With the counters picked round-robin instead, dependent groups could get neighbouring tickets and the grouped kernel's waves waited: the
 * eight-picture grid took 3.2 ms where the sorted order of round 3 took 2.5.) */
__device__ __forceinline__ uint32_t plan_cell_key(const PlanArgs &a, const uint32_t c)
{
    const int pc = c >= a.cell_off[2] && a.cgw[2] ? 2 : (c >= a.cell_off[1] && a.cgw[1] ? 1 : 0);
    const uint32_t local = c - a.cell_off[pc];
    /* (a float product, rounded however it is rounded: the stripe only has to be the SAME wherever a cell's key is worked out, and roughly its
     * place in the plane -- the 64-bit division this was cost each of the two kernels that call it a hundred instructions a cell) */
    const uint32_t stripe = min((uint32_t)((float)local * a.stripe_scale[pc]), (1u << a.shard_log2) - 1u);
    return (a.cell_depth[c] << a.shard_log2) | stripe;
}
/* One atomic add per DISTINCT counter and wave, not per cell: the lanes that share a counter add their runs up first (a row of a tile grid has
 * eight distinct depths in a wave's 64 cells; 200 000 adds on the few hundred hot words of such a grid took 0.13 ms per kernel, on two dozen
 * words 0.4 ms).  Returns, for a lane with runs, the sum of the runs of the lower lanes that share its counter; *grant (BASE only) = what the
 * counter held before the wave's add. */
template <bool BASE>
__device__ __forceinline__ uint32_t plan_wave_add(uint32_t *counters, const uint32_t key, const uint32_t nr, uint32_t *grant)
{
    const int lane = threadIdx.x & 63;
    bool todo = nr != 0;
    /* first, without touching memory: which lanes share a counter, who speaks for them (the first lane of the group), the group's total and
     * what the lanes in front of me add -- a round per DISTINCT counter of the wave.  Then ONE atomic instruction for all the groups (their
     * leaders), and the leaders' answers handed to their groups.  (With the atomic inside the loop every round was a trip to memory of its
     * own -- eight in a row for a wave of a tile grid's row, a returning one each in k_plan_cell_base: 0.13 ms next to the programs kernel.) */
    uint32_t before = 0, total = 0;
    int leader = lane;
    bool lead = false;
    for (;;) {
        const unsigned long long left = __builtin_amdgcn_ballot_w64(todo);
        if (!left) break;
        const int first = __builtin_ctzll(left);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
        const bool mine = todo && key == k0;
        if (__popcll(__builtin_amdgcn_ballot_w64(mine)) < 4) {
            /* the lanes' counters are (mostly) different ones -- a picture that is not a grid of tiles: neighbouring cells of a row have
             * different depths -- and going through them counter by counter would be up to 64 rounds (51 and 78 us for the 24 000 cells of
             * ONE 8K picture): every lane that is left speaks for itself */
            if (todo) { lead = true; leader = lane; total = nr; before = 0; todo = false; }
            break;
        }
        const uint32_t incl = wave_incl_scan(mine ? nr : 0u, lane);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (mine) { before = incl - nr; leader = first; lead = lane == first; total = tot; todo = false; }
    }
    uint32_t old = 0;
    if (lead) old = atomicAdd(counters + key, total);
    if (BASE) *grant = (uint32_t)__shfl((int)old, leader, 64);
    return before;
}
__global__ __launch_bounds__(256) void k_plan_cell_hist(PlanArgs a)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (a.result[3]) return;
    const uint32_t nr = c < a.n_cells ? a.cell_nruns[c] : 0u;
    plan_wave_add<false>(a.hist, nr ? plan_cell_key(a, c) : 0u, nr, nullptr);
}
__global__ __launch_bounds__(256) void k_plan_cell_base(PlanArgs a)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (a.result[3]) return;
    const uint32_t nr = c < a.n_cells ? a.cell_nruns[c] : 0u;
    const uint32_t key = nr ? plan_cell_key(a, c) : 0u;
    uint32_t grant = 0;
    const uint32_t before = plan_wave_add<true>(a.fill, key, nr, &grant);
    if (nr) a.cell_base[c] = a.hist_pre[key] + grant + before;
}
__global__ __launch_bounds__(256) void k_plan_rank(PlanArgs a, uint32_t m)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (a.result[1] > m) { /* more runs than windows: some window has two (k_plan_count has refused the list already) */
        if (r == 0) a.result[0] = 1;
        return;
    }
    if (r >= a.result[1]) return;
    uint32_t rank = r; /* no wavefront keys: decode order */
    if (!a.result[3]) {
        const ffhip_hevc_tu t = a.tus[a.gstart[r]];
        const uint32_t cell = a.cell_off[t.cidx] + (uint32_t)(t.y >> a.cshift[t.cidx]) * a.cgw[t.cidx] + (uint32_t)(t.x >> a.cshift[t.cidx]);
        rank = a.cell_base[cell] + (r - a.runid[a.cell_claim[cell]]); /* the cell's first run is the run of the TU that opened it */
    }
    a.rank_of[r] = rank;
}

/* The TUs of OTHER runs TU i reads, each handed to f ONCE, in the order corner, row above (left to right), column left (top to bottom);
 * returns whether every neighbour inside the TU's window belongs to its own run (then the grouped kernel may take them from its LDS tile).
 * A neighbouring TU covers a contiguous stretch of the row above or of the column to the left, so its blocks follow each other there: "same as
 * the one before" removes every repeat, the corner TU reaching into the row or the column included (it would be their first entry).  No TU can
 * be above AND left of another.
 * "Of my own run" is decided WITHOUT the neighbour's run id: a plan is only accepted when every window has ONE run (k_plan_count refuses the
 * list otherwise and the serial kernel, which takes every neighbour from memory, decodes it), TUs are aligned to their size and so lie inside
 * one window or cover whole windows -- an earlier TU is of my run exactly when its block lies in my window.  And the owners of all edge blocks
 * are loaded up front, into registers: with a conditional atomic and a record load between one block's owner and the next the loads went out one
 * at a time -- up to 33 trips to the L2 per TU, and a second one each for the run id: 0.39 ms for the 1.84 M TUs of an eight-picture grid. */
/* (Until late in round 4 every new dependency was handled where it was found -- an LDS stash store, an atomic, a counter, under a branch, at
 * each of the 33 steps: ~80 instructions a step whether or not any lane of the wave had a dependency there, 2 600 a wave.  Now the 33 steps
 * only mark which edge blocks bring a NEW dependency -- bit 0 of *mtop the corner, bits 1..16 the row above left to right, *mleft the column
 * to the left top to bottom -- and the few that do are handled afterwards, in order, by k_plan_count.) */
__device__ __forceinline__ bool plan_dep_masks(const PlanArgs &a, const uint32_t i, const ffhip_hevc_tu &t, uint32_t *mtop, uint32_t *mleft)
{
    const int c = t.cidx, n = 1 << t.log2_size, wl = a.wl[c];
    const int wx0 = (t.x >> wl) << wl, wy0 = (t.y >> wl) << wl, wsz = 1 << wl;
    const int bwc = a.bw[c];
    const int32_t *own = a.owner + a.owner_off[c];
    const int32_t *up = own + (ptrdiff_t)((t.y >> 2) - 1) * bwc + (t.x >> 2);   /* the block row above, from my first column */
    const int32_t *lf = own + (ptrdiff_t)(t.y >> 2) * bwc + (t.x >> 2) - 1;     /* the block column to the left, from my first row */
    int32_t jc = -2, jt[16], jl[16];
    if (t.flags & 1) jc = up[-1];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const bool in = 4 * k < 2 * n;
        jt[k] = (in && ((t.avail_top >> (4 * k)) & 0xf)) ? up[k] : -2;
        jl[k] = (in && ((t.avail_left >> (4 * k)) & 0xf)) ? lf[(ptrdiff_t)k * bwc] : -2;
    }
    /* inside my window?  The row above is (from the window's second row on), up to the window's right edge; the column to the left likewise */
    const bool row_in = t.y > wy0, col_in = t.x > wx0;
    const int room_x = wx0 + wsz - t.x, room_y = wy0 + wsz - t.y;
    bool ok = true;
    uint32_t mt = 0, ml = 0;
    int32_t corner = -1, last = -1;
    auto dep = [&](int32_t j, const bool inwin, uint32_t &m, const uint32_t bit) -> int32_t {
        if (j >= (int32_t)i) j = -1; /* stamped by a later TU: held older content when the sequential decoder looked */
        const bool other = j >= 0 && !inwin;
        m |= (other && j != last && j != corner) ? bit : 0u;
        last = other ? j : last;
        ok = ok && !(inwin && j < 0);
        return other ? j : -1;
    };
    if (jc != -2) corner = dep(jc, row_in && col_in, mt, 1u);
    last = -1;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (jt[k] != -2) dep(jt[k], row_in && 4 * k < room_x, mt, 2u << k);
    last = -1;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (jl[k] != -2) dep(jl[k], col_in && 4 * k < room_y, ml, 1u << k);
    *mtop = mt; *mleft = ml;
    return ok;
}

#define PLAN_WSUB 32
__global__ __launch_bounds__(256) void k_plan_count(PlanArgs a)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    __shared__ uint32_t wsum[17];
    __shared__ uint32_t blk_base;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < a.n;
    int nd = 0;
    bool ok = true;
    uint32_t mtop = 0, mleft = 0;
    ffhip_hevc_tu t = {};
    int c = 0, cx = 0, cy = 0;
    if (live) {
        t = a.tus[i];
        c = t.cidx; cx = t.x >> a.cshift[c]; cy = t.y >> a.cshift[c];
        ok = plan_dep_masks(a, i, t, &mtop, &mleft);
        nd = __popc(mtop) + __popc(mleft); /* at most 33: never more than the grouped kernel's 64 pollers */
    }
    /* room in the wait list: this block's entries together, reserved by ONE atomic add */
    uint32_t total;
    const uint32_t off = block_excl_scan((uint32_t)nd, wsum, &total);
    if (threadIdx.x == 0) {
        /* (one counter for the whole list was 7 000 returning atomics on one word, a block of 256 TUs waiting for each: the list is cut into
         * PLAN_WSUB slices, blocks take their room from slice blockIdx mod PLAN_WSUB; a slice that runs over refuses the plan like the whole
         * list running over did) */
        const uint32_t r = blockIdx.x & (a.wsub_n - 1), slice = a.wait_cap / a.wsub_n;
        uint32_t got = 0;
        if (total) {
            got = atomicAdd(a.wsub + 32 * r, total);
            if (got + total > slice) { a.result[0] = 1; got = 0; }
        }
        blk_base = r * slice + got;
    }
    __syncthreads();
    if (!live) return;
    const uint32_t wb = blk_base + off;
    a.wbegin[i] = wb;
    a.wcount[i] = (uint32_t)nd;
    { /* the marked edge blocks, in order (corner, row above, column to the left): who owns it -- read again, the gather's registers cannot be
         indexed by a lane's own bit -- must publish a done flag, and goes into my wait list */
        const int32_t *own = a.owner + a.owner_off[c];
        const int bwc = a.bw[c];
        uint32_t q = 0, mt = mtop, ml = mleft;
        while (mt | ml) {
            int px = t.x - 1, py = t.y - 1;
            if (mt) {
                const int b = __builtin_ctz(mt);
                mt &= mt - 1;
                if (b) px = t.x + 4 * (b - 1);
            } else {
                const int b = __builtin_ctz(ml);
                ml &= ml - 1;
                py = t.y + 4 * b;
            }
            const uint32_t j = (uint32_t)own[(ptrdiff_t)(py >> 2) * bwc + (px >> 2)];
            atomicOr((unsigned *)(a.flags + (j & ~3u)), 1u << (8 * (j & 3))); /* that TU must publish a done flag */
            if (wb + q < a.wait_cap) a.wait_idx[wb + q] = j;
            q++;
        }
    }
    atomicOr((unsigned *)(a.flags + (i & ~3u)), (ok ? 2u : 0u) << (8 * (i & 3)));
    const uint32_t cell = a.cell_off[c] + (uint32_t)cy * a.cgw[c] + (uint32_t)cx;
    const bool starts = i == 0 || a.runid[i] != a.runid[i - 1];
    if (starts) {
        a.gstart[a.runid[i]] = i;
        atomicAdd(a.cell_nruns + cell, 1u);
        if (atomicCAS(a.win_run + win_of(a, t), ~0u, a.runid[i]) != ~0u) a.result[0] = 1; /* a window with two runs */
    }
    if (i == a.n - 1) {
        a.gstart[a.runid[i] + 1] = a.n;
        a.result[1] = a.runid[i] + 1;
        a.result[5] = (uint32_t)a.wl[0]; /* (diagnostics: the luma window the schedule was built for) */
    }
}

__global__ __launch_bounds__(256) void k_plan_emit(PlanArgs a, uint32_t m)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    if (a.result[1] > m) return; /* refused (k_plan_rank): the runs have no tickets, and nobody will read a schedule */
    const ffhip_hevc_tu t = a.tus[i];
    const uint32_t wb = a.wbegin[i], wc = a.wcount[i];
    if (wc) {
        const uint32_t my_ticket = a.rank_of[a.runid[i]];
        for (uint32_t q = 0; q < wc && wb + q < a.wait_cap; q++)
            if (a.rank_of[a.runid[a.wait_idx[wb + q]]] >= my_ticket) a.result[0] = 1; /* would wait for a later ticket: not with this order */
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
    /* (the second quarter belongs to k_hevc_intra_program, which writes it for EVERY slot and may be running next to this kernel: the
     * availability masks that once sat there are in the substitution table) */
    a.sched[(size_t)i * 3 + 2] = q2;
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

__global__ __launch_bounds__(256) void k_plan_runid(PlanArgs a)
{
    if (a.result[6]) return; /* the list was refused by k_hevc_check_tus */
    /* run id = starts in the blocks before (scanned totals) + starts up to and including me in this block - 1 */
    __shared__ uint32_t wtot[4];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool st = i < a.n && a.start[i] != 0;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(st);
    if (lane == 0) wtot[w] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t before = a.blk_pre[blockIdx.x];
    for (int k = 0; k < w; k++) before += wtot[k];
    const uint32_t incl = (uint32_t)__popcll(b & ((2ull << lane) - 1ull));
    if (i < a.n) a.runid[i] = before + incl - 1u;
}

/* Validation of a large TU list ON the device (what ffhip_hevc_intra_recon's host pass checks record by record: field ranges, the block
 * inside its plane, no availability bit pointing outside the plane, a residual buffer where a TU asks for one).  For lists of 2^17 TUs and
 * more the host only looks at a sample (a pass over 1.8 M records was 0.7 ms of an enqueue call next to 4 ms of device work); a bad record
 * found here refuses the call through the stream: result[6] sends every kernel behind this one home, result[0] the grouped kernel, and
 * ffhip_stream_sync reports FFHIP_EINVAL -- nothing is written. */
struct PlanCheck {
    int pw[3], ph[3];
    int chroma_ok;    /* chroma planes present and wide enough */
    int have_residual;
    int *async_err;
};
__global__ __launch_bounds__(256) void k_hevc_check_tus(const ffhip_hevc_tu *tus, uint32_t n, PlanCheck k, uint32_t *result)
{
    bool bad = false;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const ffhip_hevc_tu t = tus[i];
        const int c = t.cidx;
        if (c > 2 || t.log2_size < 2 || t.log2_size > 5 || t.pred_mode > 34) { bad = true; continue; }
        const int sz = 1 << t.log2_size;
        if (c > 0 && !k.chroma_ok) bad = true;
        if (t.x + sz > k.pw[c] || t.y + sz > k.ph[c]) { bad = true; continue; }
        const unsigned long long span = sz == 32 ? ~0ull : (1ull << (2 * sz)) - 1;
        const unsigned long long top = t.avail_top & span, left = t.avail_left & span;
        const int room_x = k.pw[c] - t.x, room_y = k.ph[c] - t.y;
        if ((top || (t.flags & 1)) && t.y == 0) bad = true;
        if ((left || (t.flags & 1)) && t.x == 0) bad = true;
        if (room_x < 64 && (top >> room_x)) bad = true;
        if (room_y < 64 && (left >> room_y)) bad = true;
        if ((t.flags & 2) && !k.have_residual) bad = true;
    }
    if (__builtin_amdgcn_ballot_w64(bad) && (threadIdx.x & 63) == 0) {
        result[6] = 1u; result[0] = 1u; result[3] = 1u;
        __hip_atomic_store(k.async_err, FFHIP_ASYNC_BAD_INPUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

/* Layout of the device scratch the caller provides (32-bit words).  sched / groups / wait_idx sit where the grouped
 * kernel expects to be told they are; everything else is planner-private.  ONE function lays the scratch out, for the size
 * query (base = NULL: only the word count matters) and for the launch. */
struct PlanLayout {
    size_t words, blocks, wins, cells, n_blocks, wait_cap;
    uint32_t *zero_cells; /* cell_edges | cell_nruns | hist | fill, adjacent: cleared together */
    size_t zero_cells_words;
};
static PlanLayout plan_layout(PlanArgs &a, uint32_t *base, const ffhip_hevc_tu *d_tus, long long n_tus, const int pw[3], const int ph[3], const int wl[3])
{
    PlanLayout L = {};
    const size_t n = (size_t)n_tus;
    a.tus = d_tus;
    a.n = (uint32_t)n;
    size_t blocks = 0, wins = 0, cells = 0;
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
        a.cshift[c] = c == 0 ? 6 : 6 - ((pw[c] > 0 && pw[c] * 2 <= pw[0] + 1) ? 1 : 0); /* the cell is 64x64 LUMA samples */
        a.cgw[c] = pw[c] > 0 ? (uint32_t)(((pw[c] - 1) >> a.cshift[c]) + 1) : 0;
        a.cgh[c] = pw[c] > 0 ? (uint32_t)(((ph[c] - 1) >> a.cshift[c]) + 1) : 0;
        a.cell_off[c] = (uint32_t)cells;
        if (pw[c] > 0) cells += (size_t)a.cgw[c] * (size_t)a.cgh[c];
    }
    a.n_cells = (uint32_t)cells;
    L.blocks = blocks; L.wins = wins; L.cells = cells;
    L.n_blocks = (n + 255) / 256;
    L.wait_cap = 8 * n; /* 66 entries per TU bounds the list in theory; 8n is reserved, and a list that needs more is refused (result[2]) */
    a.wait_cap = (uint32_t)L.wait_cap;
    uint32_t *p = base;
    a.sched = (u32x4 *)p; p += 12 * n;
    a.groups = (u32x4 *)p; p += 4 * (n + 1);
    a.wait_idx = p; p += L.wait_cap;
    a.owner = (int32_t *)p; p += blocks;          /* owner | win_run: adjacent, set to ~0 together */
    a.win_run = p; p += wins;
    a.start = p; p += n;
    a.runid = p; p += n;
    a.wcount = p; p += n;
    a.wbegin = p; p += n;
    a.gstart = p; p += n + 2;
    a.flags = (uint8_t *)p; p += (n + 3) / 4 + 4;   /* flags | result: adjacent, cleared together */
    a.result = p; p += 16;
    a.wsub_n = 1; while (a.wsub_n < PLAN_WSUB && 2 * (size_t)a.wsub_n <= L.n_blocks) a.wsub_n *= 2;
    a.wsub = p ? p + ((32 - (((uintptr_t)p >> 2) & 31)) & 31) : p; p += 32 * PLAN_WSUB + 32; /* (cleared with flags | result) */
    a.cell_claim = p; p += cells;
    L.zero_cells = p;
    a.cell_edges = p; p += cells;
    a.cell_nruns = p; p += cells;
    uint32_t depths = 1;
    for (int c = 0; c < 3; c++) depths = a.cgw[c] + 2 * a.cgh[c] + 1 > depths ? a.cgw[c] + 2 * a.cgh[c] + 1 : depths;
    a.depths = depths;
    a.shard_log2 = 0; /* as many shards as keep the table at 32 K words or below, at most 64 */
    while (a.shard_log2 < 6 && ((size_t)depths << (a.shard_log2 + 1)) <= 32768) a.shard_log2++;
    for (int c = 0; c < 3; c++) a.stripe_scale[c] = a.cgw[c] * a.cgh[c] ? (float)(1u << a.shard_log2) / (float)(a.cgw[c] * a.cgh[c]) : 0.0f;
    const size_t hwords = (size_t)depths << a.shard_log2;
    a.hist = p; p += hwords;
    a.fill = p; p += hwords;
    L.zero_cells_words = 2 * cells + 2 * hwords;
    a.cell_depth = p; p += cells;
    a.cell_base = p; p += cells;
    a.rank_of = p; p += n;
    a.blk_tot = p; p += L.n_blocks + 1;
    a.blk_pre = p; p += L.n_blocks + 1;
    a.hist_pre = p; p += hwords;
    a.part_nb = (uint32_t)L.n_blocks;
    a.part_tot = p; p += 3 * L.n_blocks + 1;
    a.part_pre = p; p += 3 * L.n_blocks + 1;
    p += (8 - (((uintptr_t)p >> 2) & 7)) & 7; /* records are read and written as 16-byte quarters */
    a.sorted = (ffhip_hevc_tu *)p; p += 8 * n;
    a.raw = d_tus;
    L.words = (size_t)(p - base) + 16;
    return L;
}
extern "C" size_t ffhip_hevc_plan_gpu_words(long long n_tus, const int pw[3], const int ph[3], const int wl[3])
{
    PlanArgs a;
    return plan_layout(a, nullptr, nullptr, n_tus, pw, ph, wl).words;
}

/* Returns 0 when the plan is in place (n_groups, n_wait filled), 1 when the list needs the host planner.
 * With d_result != NULL nothing is waited for: the plan is only ENQUEUED, *d_result points at the device words
 * {refused, number of groups, wait entries} the grouped kernel reads for itself (with *wait_cap, the reservation the
 * wait entries must fit), *n_groups is left alone and the return value is 0. */
extern "C" int ffhip_hevc_plan_gpu_checked(const ffhip_hevc_tu *d_tus, long long n_tus, const int pw[3], const int ph[3], const int wl[3],
                                           uint32_t *scratch, hipStream_t st, const u32x4 **sched, const u32x4 **groups, const uint32_t **wait_idx,
                                           int *n_groups, const uint32_t **d_result, uint32_t *wait_cap_out, const int *check /* NULL, or {chroma_ok,
                                           have_residual} */, int *async_err, const FfhipPlanHooks *hooks, uint32_t *also_zero, size_t also_zero_words);
extern "C" int ffhip_hevc_plan_gpu(const ffhip_hevc_tu *d_tus, long long n_tus, const int pw[3], const int ph[3], const int wl[3],
                                   uint32_t *scratch, hipStream_t st, const u32x4 **sched, const u32x4 **groups, const uint32_t **wait_idx,
                                   int *n_groups, const uint32_t **d_result, uint32_t *wait_cap_out)
{
    return ffhip_hevc_plan_gpu_checked(d_tus, n_tus, pw, ph, wl, scratch, st, sched, groups, wait_idx, n_groups, d_result, wait_cap_out, nullptr, nullptr,
                                       nullptr, nullptr, 0);
}
/* ... with the list's validation as the first kernel behind the scratch's reset (check != NULL), a hook that runs once that kernel is
 * enqueued: what the caller starts from there (the substitution table on a side stream) may rely on result[6], handed to the hook; and a
 * second hook behind k_plan_count, when the per-TU flags (who publishes, who may use the LDS tile) and wait counts are final: the per-pixel
 * programs need nothing else of the schedule and can be built next to the ticket kernels (flags, wait counts, result words) */
extern "C" int ffhip_hevc_plan_gpu_checked(const ffhip_hevc_tu *d_tus, long long n_tus, const int pw[3], const int ph[3], const int wl[3],
                                           uint32_t *scratch, hipStream_t st, const u32x4 **sched, const u32x4 **groups, const uint32_t **wait_idx,
                                           int *n_groups, const uint32_t **d_result, uint32_t *wait_cap_out, const int *check, int *async_err,
                                           const FfhipPlanHooks *hooks, uint32_t *also_zero /* a region of the
                                           caller's (the grouped kernel's ticket counter and done flags), cleared by the same launch */, size_t also_zero_words)
{
    PlanArgs a;
    const PlanLayout Lo = plan_layout(a, scratch, d_tus, n_tus, pw, ph, wl);
    const size_t n = (size_t)n_tus, blocks = Lo.blocks, wins = Lo.wins, cells = Lo.cells, wait_cap = Lo.wait_cap;
    /* owner = -1, win_run = ~0 (adjacent); flags, result = 0 (adjacent); cell_claim = ~0; cell_edges, cell_nruns, hist, fill = 0 (adjacent:
     * OR-ed and added into; every cell's depth is written by the sweep): ONE launch for the four regions -- as four memsets they were four
     * more kernel boundaries in front of a chain of small kernels */
    {
        PlanInit in;
        in.p[0] = (uint32_t *)a.owner; in.words[0] = blocks + wins; in.value[0] = ~0u;
        in.p[1] = (uint32_t *)a.flags; in.words[1] = (n + 3) / 4 + 4 + 16 + 32 * PLAN_WSUB + 32; in.value[1] = 0u;
        in.p[2] = a.cell_claim; in.words[2] = cells; in.value[2] = ~0u;
        in.p[3] = Lo.zero_cells; in.words[3] = Lo.zero_cells_words; in.value[3] = 0u;
        in.p[4] = also_zero; in.words[4] = also_zero ? also_zero_words : 0; in.value[4] = 0u;
        size_t most = 0;
        for (int r = 0; r < 5; r++) most = in.words[r] > most ? in.words[r] : most;
        const size_t wg = (most / 4 + 255) / 256 + 1;
        hipLaunchKernelGGL(k_plan_init, dim3((unsigned)(wg > 4096 ? 4096 : wg), also_zero ? 5 : 4), dim3(256), 0, st, in);
    }
    const unsigned grid = (unsigned)Lo.n_blocks;
    if (check && async_err) {
        PlanCheck k;
        for (int c = 0; c < 3; c++) { k.pw[c] = pw[c]; k.ph[c] = ph[c]; }
        k.chroma_ok = check[0]; k.have_residual = check[1]; k.async_err = async_err;
        hipLaunchKernelGGL(k_hevc_check_tus, dim3(grid > 2048 ? 2048u : grid), dim3(256), 0, st, d_tus, (uint32_t)n, k, a.result);
    }
    if (hooks && hooks->after_check) {
        const int hrc = hooks->after_check(hooks->ctx, a.result + 6);
        if (hrc) return hrc;
    }
    if (hooks && hooks->by_plane) { /* the list sorted by plane (k_part_*): from here on `a.tus` is the sorted copy */
        hipLaunchKernelGGL(k_part_count, dim3(grid), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_plan_scan, dim3((unsigned)((3 * Lo.n_blocks + 4095) / 4096)), dim3(256), 0, st, (const uint32_t *)a.part_tot, a.part_pre, (uint32_t)(3 * Lo.n_blocks),
                           (uint32_t *)nullptr, 0u, (const uint32_t *)(a.result + 6));
        hipLaunchKernelGGL(k_part_scatter, dim3(grid), dim3(256), 0, st, a);
        a.tus = a.sorted;
    }
    if (hooks && hooks->tus_used) *hooks->tus_used = a.tus;
    hipLaunchKernelGGL(k_plan_owner, dim3(grid), dim3(256), 0, st, a);
    /* the depth sweep needs the cells' edges, which k_plan_owner has just left: three workgroups walking diagonals for 50 - 130 us -- next to
     * k_plan_count on a stream of the caller's where there is one, in front of the ticket kernels otherwise */
    auto enqueue_sweep = [&](hipStream_t ss) {
        uint32_t max_gh = 0;
        for (int c = 0; c < 3; c++) max_gh = a.cgh[c] > max_gh ? a.cgh[c] : max_gh;
        const unsigned threads = max_gh >= 1024 ? 1024u : (unsigned)((max_gh + 63) / 64 * 64);
        bool fast = max_gh <= DEPTH_ROWS;
        for (int c = 0; c < 3; c++) fast = fast && (size_t)a.cgw[c] * a.cgh[c] <= CELLS_LDS;
        uint32_t max_gw = 0;
        for (int c = 0; c < 3; c++) max_gw = a.cgw[c] > max_gw ? a.cgw[c] : max_gw;
        if (max_gw <= 1024 && fast && !FFHIP_ENV("FFHIP_HEVC_DEPTH_DIAGONALS")) /* row by row: planes of up to 65 536 samples a line whose edge bits fit the LDS */
            hipLaunchKernelGGL(k_plan_cell_depth_rows, dim3(3), dim3(1024), 0, ss, a); /* (1024 threads fetch the edge bits; the waves without columns leave then) */
        else if (fast) hipLaunchKernelGGL(k_plan_cell_depth<true>, dim3(3), dim3(1024), 0, ss, a);
        else hipLaunchKernelGGL(k_plan_cell_depth<false>, dim3(3), dim3(threads ? threads : 64u), 0, ss, a);
    };
    hipStream_t ts = st; /* the stream of the sweep, the ticket kernels and k_plan_emit */
    if (hooks && hooks->ticket_stream) {
        void *ss = hooks->ticket_stream(hooks->ctx);
        if (ss) {
            ts = (hipStream_t)ss;
            enqueue_sweep(ts);
        }
    }
    hipLaunchKernelGGL(k_plan_scan, dim3((unsigned)((Lo.n_blocks + 4095) / 4096)), dim3(256), 0, st, (const uint32_t *)a.blk_tot, a.blk_pre, (uint32_t)Lo.n_blocks, (uint32_t *)nullptr, 0u,
                       (const uint32_t *)(a.result + 6));
    hipLaunchKernelGGL(k_plan_runid, dim3(grid), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_plan_count, dim3(grid), dim3(256), 0, st, a);
    if (hooks && hooks->after_count) {
        const int hrc = hooks->after_count(hooks->ctx, a.flags, a.wcount, a.result);
        if (hrc) return hrc;
    }
    /* tickets: runs by (wavefront index of their cell, decode order).  The sweep: one diagonal per step, at most one cell per row of cells, a
     * lane per row (k_plan_cell_depth) */
    if (ts == st) enqueue_sweep(st);
    else if (hooks->tickets_wait) { const int hrc = hooks->tickets_wait(hooks->ctx); if (hrc) return hrc; }
    const size_t m = n < wins ? n : wins; /* runs <= windows, or the plan is refused (k_plan_count: a window with two runs) */
    const unsigned cgrid = (unsigned)((cells + 255) / 256);
    hipLaunchKernelGGL(k_plan_cell_hist, dim3(cgrid), dim3(256), 0, ts, a);
    hipLaunchKernelGGL(k_plan_scan, dim3((unsigned)((((size_t)a.depths << a.shard_log2) + 4095) / 4096)), dim3(256), 0, ts, (const uint32_t *)a.hist, a.hist_pre,
                       (uint32_t)(a.depths << a.shard_log2), a.result + 4, a.shard_log2, (const uint32_t *)(a.result + 3));
    hipLaunchKernelGGL(k_plan_cell_base, dim3(cgrid), dim3(256), 0, ts, a);
    hipLaunchKernelGGL(k_plan_rank, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ts, a, (uint32_t)m);
    hipLaunchKernelGGL(k_plan_emit, dim3(grid), dim3(256), 0, ts, a, (uint32_t)m);
    if (ts != st && hooks->tickets_enqueued) { const int hrc = hooks->tickets_enqueued(hooks->ctx); if (hrc) return hrc; }
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
