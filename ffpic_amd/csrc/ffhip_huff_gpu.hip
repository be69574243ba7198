/*
 * ffhip_huff_gpu.hip -- the JPEG entropy front end ON the GPU for files that carry restart markers
 * ("Huffman is serial per restart interval" -- so the intervals are the parallelism).
 *
 * What the host still does per file: the marker loop and table building of ffhip_entropy.c
 * (format/jpg.c:78-105, 640-655, 771-855) and a scan for the RSTn markers.  What moves to the device:
 * read_compressed_scan / decode_data_unit (format/jpg.c:255-415, 562-573, 588-637) -- ONE LANE per
 * restart interval walks its bytes (unstuffed by the host while it stages them; 64-bit bit buffer), decodes symbol after symbol with
 * the same 9-bit look-up tables the host decoder uses (struct huff, uploaded as is) and scatters the
 * coefficients, de-zigzagged, into the MCU-order planes ffhip_jpeg_recon_batch reads.  The decode loop is
 * a flat one-symbol-per-iteration state machine so that lanes in different blocks, components or MCUs
 * still execute the same instructions.  A file without a DRI segment is one interval = one lane: accepted,
 * but only a large batch of such files fills the machine (the callers in ffhip_pipeline.hip send them to the
 * host threads unless there are a thousand or more).
 */
#include "ffhip_internal.h"
#include <chrono>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "ffhip_entropy_internal.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>

struct HuffImage {
    uint32_t scan_off, scan_len; /* this picture's entropy-coded bytes inside `scan` */
    uint32_t seg_base, n_seg;    /* its restart intervals inside `seg` (offsets relative to scan_off): n_seg starts and,
                                  * as entry n_seg, the length of the clean stream = the end of the last interval */
    uint32_t restart, mcus;
    uint32_t ncomp, nb[3];       /* blocks per MCU and component */
    uint32_t tab_dc[3], tab_ac[3]; /* indices into `tabs` */
};

struct HuffArgs {
    const uint8_t *scan;
    const struct huff *tabs;
    const uint16_t *lut; /* per table of `tabs`: LUT_WORDS entries, see build_lut */
    const HuffImage *images;
    const uint32_t *seg;
    const u32x2 *work; /* (picture, interval) per lane */
    int16_t *plane[3];
    int *status;       /* per picture, device: non-zero if some interval was malformed */
    uint32_t n_work;
};

__device__ static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                               12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                               58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* The byte streams the kernel reads are UNSTUFFED (the host removes the 00 behind every FF while it stages the
 * bytes) and every restart interval starts 4-byte aligned and is followed by zero padding -- what the reference's
 * reader feeds itself at a marker.  Why the shape below: lanes of a wave advance independently, so any
 * memory wait inside a data-dependent branch is paid by the whole wave on almost every iteration (with 64 lanes
 * SOME lane always needs bytes, SOME lane always ends a block).  Hence
 *  - bytes come from a 128-byte ring per lane in LDS ([dword][lane]: conflict-free; 256 bytes until round 5), topped up for all lanes at
 *    once every 16 symbols by aligned 16-byte global loads -- one memory wait per 16 iterations per wave;
 *  - the next dword of the ring is read at the top of every iteration, needed or not, and spliced into the
 *    64-bit bit buffer at the bottom: the LDS latency hides behind the symbol decode;
 *  - what changes per block only (block base, component tables, predictor) is state, not recomputed per symbol. */
#define LUT_GROUPS 8
#define LUT_WORDS (512 + 128 * LUT_GROUPS) /* 1536 uint16 = 3 KB per table */

template <int RING_DW, int REFILL_EVERY> /* dwords of a lane's byte ring, symbols between two refills (at most 31 bits a symbol: RING_DW >= REFILL_EVERY + 8) */
__global__ __launch_bounds__(256) void k_jpeg_huff(HuffArgs a)
{
    __shared__ uint8_t zz[64];
    /* two-level look-up tables of the wave's FIRST picture in LDS (a wave's 64 intervals belong to one picture, rarely
     * two): [0..2] DC, [3..5] AC, per component.  Two levels, not the host's 9-bit table plus canonical-code walk: a
     * code that misses sends its lane into a loop of loads from global memory, and with 64 lanes per wave the 2-5 %
     * of symbols with long codes (JPEG's 16-bit codes mostly) mean the whole wave walks on nearly every symbol.
     * (A 12-bit single level does not help -- the misses ARE the 15/16-bit codes -- and costs 48 KB of LDS; the host
     * decoder's one-look-up run/value path for small AC coefficients lost for the same serialisation reason.) */
    /* FOUR waves to a workgroup (round 5), sharing the tables: with a wave per workgroup the kernel's 38 KB of LDS per wave -- these 18 KB of tables, a
     * 16 KB ring, 4 KB of block-change state -- allowed ONE wave per SIMD, and a batch of more than 1 024 waves (486 4K files of 135 intervals) ran in
     * rounds: 1 024 files 32.2 ms where 256 take 12.6.  Shared tables and half the ring are 17 KB a wave: two waves per SIMD. */
    __shared__ uint16_t lt[6][LUT_WORDS];
    __shared__ uint32_t ring[4][RING_DW][64];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 64) zz[threadIdx.x] = kZigzag[threadIdx.x];
    const uint32_t gid0 = blockIdx.x * 256;
    const uint32_t img0 = a.work[gid0].x; /* gid0 < n_work: the grid is not larger than the work list */
    const HuffImage im0 = a.images[img0];
    for (int t = 0; t < 3; t++) {
        const u32x4 *sd = (const u32x4 *)(a.lut + (size_t)im0.tab_dc[t] * LUT_WORDS), *sa = (const u32x4 *)(a.lut + (size_t)im0.tab_ac[t] * LUT_WORDS);
        for (int i = (int)threadIdx.x; i < LUT_WORDS / 8; i += 256) {
            ((u32x4 *)lt[t])[i] = sd[i];
            ((u32x4 *)lt[3 + t])[i] = sa[i];
        }
    }
    __syncthreads(); /* (the kernel's only barrier: every wave of the workgroup reaches it, whether or not it has intervals) */
    const uint32_t gid = gid0 + wv * 64 + lane;
    const bool exists = gid < a.n_work;
    const u32x2 w = a.work[exists ? gid : gid0];
    const HuffImage im = a.images[w.x];
    /* the tables in LDS are the first interval's picture's: good for every picture that uses the SAME tables (distinct tables are uploaded once per
     * batch and pictures refer to them by index: an encoder's defaults are one set for the whole batch) */
    bool in_lds = true;
    for (int t = 0; t < 3; t++) in_lds = in_lds && im.tab_dc[t] == im0.tab_dc[t] && im.tab_ac[t] == im0.tab_ac[t];
    /* this lane's byte stream: 16-byte aligned chunks from `src`, and never a byte that is not the interval's own:
     * dwords at or behind `dw_end` (the end of the interval with its zero padding = the start of the next one) read as
     * zero, as the host reader feeds itself zeros at `end`, and a lane that has taken more than the look-ahead of a
     * well-formed stream from there is malformed (truncated file, empty interval, DRI larger than the data) and stops */
    const uint32_t start = im.scan_off + a.seg[im.seg_base + w.y]; /* 4-byte aligned */
    const u32x4 *src = (const u32x4 *)(a.scan + (start & ~15u));
    uint32_t rd = (start & 15u) >> 2, wr = 0; /* dword cursors into the stream counted from src */
    const uint32_t dw_end = (im.scan_off + a.seg[im.seg_base + w.y + 1] - (start & ~15u)) >> 2; /* seg[n_seg] = clean length */
    uint32_t mcu = w.y * im.restart;
    const uint32_t mcu_end = mcu + im.restart < im.mcus ? mcu + im.restart : im.mcus;
    /* ---- what a block change needs, as per-lane tables in LDS, so that it is a few look-ups and not chains of
     * three-way selects: per block slot of the MCU its component, tables and block count; per component the
     * predictor and the byte address of the picture's plane ---- */
    __shared__ uint32_t slotrec[4][8][64];         /* c | kb << 2 | nb_c << 5 | tix_dc << 8 | tix_ac << 20 */
    __shared__ int predv[4][3][64];
    __shared__ unsigned long long planeb[4][3][64];
    uint32_t nbt = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const unsigned long long pb = a.plane[c] ? (unsigned long long)(uintptr_t)(a.plane[c] + (size_t)w.x * im.mcus * im.nb[c] * 64) : 0ull;
        planeb[wv][c][lane] = pb;
        predv[wv][c][lane] = 0;
        for (uint32_t kb0 = 0; kb0 < im.nb[c] && nbt < 8; kb0++)
            slotrec[wv][nbt++][lane] = (uint32_t)c | (kb0 << 2) | (im.nb[c] << 5) | ((im.tab_dc[c] & 0xfffu) << 8) | ((im.tab_ac[c] & 0xfffu) << 20);
    }
    unsigned long long acc = 0; /* LEFT-aligned: next unread bit is bit 63 */
    int n = 0;
    bool active = exists && mcu < mcu_end, bad = false;
    uint32_t slot = 0, k = 0, rec = slotrec[wv][0][lane];
    int16_t *blk = (int16_t *)(uintptr_t)planeb[wv][0][lane] + (size_t)mcu * im.nb[0] * 64;
    int pred_cur = 0;
    auto refill = [&]() { /* all lanes: fetch 16-byte chunks while the ring has room for one */
        while (__builtin_amdgcn_ballot_w64(wr + 4 <= rd + RING_DW)) {
            if (wr + 4 <= rd + RING_DW) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (wr < dw_end) v = src[wr >> 2]; /* a chunk that starts inside the interval ends inside the staged buffer */
#pragma unroll
                for (int j = 0; j < 4; j++) ring[wv][(wr + j) & (RING_DW - 1)][lane] = wr + j < dw_end ? v[j] : 0u;
                wr += 4;
            }
        }
    };
    refill();
    for (int i = 0; i < 2; i++) { /* 64 bits to start with */
        const uint32_t d = ring[wv][rd & (RING_DW - 1)][lane];
        acc = (acc << 32) | __builtin_bswap32(d);
        rd++;
    }
    n = 64;
    for (uint32_t iter = 1; __builtin_amdgcn_ballot_w64(active); iter++) {
        if ((iter & (REFILL_EVERY - 1)) == 0) refill(); /* at most REFILL_EVERY x 31 bits < REFILL_EVERY dwords used since the last one */
        const uint32_t nextdw = ring[wv][rd & (RING_DW - 1)][lane]; /* wanted at the bottom, if at all */
        if (active) {
            const bool dc = k == 0;
            const uint32_t c = rec & 3u;
            /* one Huffman symbol (coding/huffman.c:92-222): two-level look-up, canonical-code walk as the last resort */
            const unsigned top = (unsigned)(acc >> 32);
            const unsigned peek = top >> (32 - LOOK);
            const uint32_t tix = dc ? (rec >> 8) & 0xfffu : rec >> 20;
            unsigned e;
            if (in_lds) { /* two separate paths, not one pointer into either memory: that would be a flat load per symbol */
                const uint16_t *l = lt[dc ? c : 3 + c];
                e = l[peek];
                if (e & 0x8000u) e = l[512 + ((e & 0xffu) << 7) + ((top >> (32 - 16)) & 127u)]; /* long code: its group, next 7 bits */
            } else {
                const uint16_t *l = a.lut + (size_t)tix * LUT_WORDS;
                e = l[peek];
                if (e & 0x8000u) e = l[512 + ((e & 0xffu) << 7) + ((top >> (32 - 16)) & 127u)];
            }
            int sym = (int)(e & 0xff), len = (int)(e >> 8);
            if (!e) { /* a table with more long-code groups than the LUT holds, or a code that does not exist */
                const struct huff *T = a.tabs + tix;
                int code = (int)peek;
                len = LOOK;
                while (len < 17 && code > T->maxcode[len]) {
                    len++;
                    code = (int)(top >> (32 - len));
                }
                if (len > 16) { bad = true; len = 16; }
                sym = T->vals[(T->valptr[len] + code - T->mincode[len]) & 255]; /* & 255: a malformed DHT must not index outside the table */
            }
            acc <<= len;
            /* ---- the rest of the step without branches: run, magnitude bits, EXTEND, predictor, store ---- */
            const int s = dc ? sym : (sym & 15), r = dc ? 0 : (sym >> 4);
            const bool skip = !dc && s == 0;                       /* ZRL or EOB: nothing to store */
            k += skip ? (r == 15 ? 16u : 64u) : (uint32_t)r;       /* EOB: anything that ends the block */
            const unsigned vb = (unsigned)((acc >> 33) >> (31 - s)); /* the next s bits; s = 0 gives 0 */
            acc <<= s;
            n -= len + s;
            const int lim = (1 << s) >> 1;                         /* 1 << (s - 1), 0 for s = 0 */
            int v = (int)vb < lim ? (int)vb - (1 << s) + 1 : (int)vb; /* EXTEND, T.81 F.2.2.1 */
            pred_cur += dc ? v : 0;
            v = dc ? pred_cur : v;
            bad |= (dc && s > 11) || (!skip && k > 63);
            if (!skip && k <= 63) blk[zz[k]] = (int16_t)v;
            k += skip ? 0u : 1u;
            if (k >= 64) { /* next block: the MCU's next slot, or the next MCU */
                k = 0;
                slot++;
                const bool wrap = slot == nbt;
                slot = wrap ? 0u : slot;
                mcu += wrap ? 1u : 0u;
                rec = slotrec[wv][slot][lane];
                const uint32_t c2 = rec & 3u, nbc = (rec >> 5) & 7u, kb = (rec >> 2) & 7u;
                predv[wv][c][lane] = pred_cur;      /* LDS keeps program order: when c2 == c the read returns this value */
                pred_cur = predv[wv][c2][lane];
                blk = (int16_t *)(uintptr_t)planeb[wv][c2][lane] + (size_t)(mcu * nbc + kb) * 64;
                if (mcu >= mcu_end) active = false;
            }
            /* a well-formed interval ends before its padding: rd is at most two dwords ahead of the bits consumed */
            bad |= rd > dw_end + 2;
#ifdef HUFF_DEBUG
            if (rd > dw_end + 2) printf("lane %u img %u seg %u: rd %u lim %u start %u mcu %u/%u k %u n %d\n", gid, w.x, w.y, rd, dw_end, start, mcu, mcu_end, k, n);
#endif
            if (bad) active = false;
            if (n <= 32) { /* splice the dword read at the top behind the n valid bits */
                acc |= (unsigned long long)__builtin_bswap32(nextdw) << (32 - n);
                n += 32;
                rd++;
            }
        }
    }
    if (bad) a.status[w.x] = FFHIP_EINVAL;
}

#define SCRATCH_HUFF 4

namespace {
/* One pass over a picture's entropy-coded segment: the bytes without their stuffing (FF 00 -> FF) into dst, every
 * restart interval 4-byte aligned and followed by at least 4 zero bytes, seg[k] = offset of interval k in that clean
 * stream.  RSTn markers separate the intervals; any other marker (or the n_seg-th RSTn) ends the scan.  Returns the
 * number of intervals found; *clean_len = bytes written.  16 bytes at a time while no 0xFF is in sight (one in 256
 * bytes is, in entropy-coded data): the first form, memchr + memcpy per run and a separate marker scan before it,
 * spent its time in call overhead -- 7 ms of staging per 256 4K files on 16 threads, against 12 ms of kernel.
 * May store up to 15 bytes past the clean length: the caller reserves the slack. */
uint32_t stage_scan(uint8_t *dst, const uint8_t *src, const uint8_t *end, uint32_t *seg, uint32_t n_seg, size_t *clean_len)
{
    uint8_t *d = dst;
    uint32_t k = 0;
    seg[0] = 0;
    for (;;) {
#if defined(__SSE2__)
        const __m128i ff = _mm_set1_epi8((char)0xFF);
        while (src + 16 <= end) {
            const __m128i v = _mm_loadu_si128((const __m128i *)src);
            const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, ff));
            _mm_storeu_si128((__m128i *)d, v);
            if (m) {
                const int c = __builtin_ctz(m);
                src += c;
                d += c;
                break;
            }
            src += 16;
            d += 16;
        }
#endif
        while (src < end && *src != 0xFF) *d++ = *src++;
        if (src + 1 >= end) break; /* end of data (a lone trailing FF is not data) */
        const uint8_t b = src[1];
        if (b == 0) { *d++ = 0xFF; src += 2; continue; }
        if (b < 0xD0 || b > 0xD7 || k + 1 >= n_seg) break; /* another marker, or more RSTn than intervals: the scan ends */
        {
            const size_t len = (size_t)(d - dst) - seg[k], padded = ((len + 3) & ~(size_t)3) + 4;
            memset(d, 0, padded - len);
            d = dst + seg[k] + padded;
            seg[++k] = (uint32_t)(d - dst);
            src += 2;
        }
    }
    {
        const size_t len = (size_t)(d - dst) - seg[k], padded = ((len + 3) & ~(size_t)3) + 4;
        memset(d, 0, padded - len);
        d = dst + seg[k] + padded;
    }
    *clean_len = (size_t)(d - dst);
    return k + 1;
}

/* Two-level table of a canonical Huffman code, LUT_WORDS uint16:
 *   [0..511]            by the first 9 bits: (length << 8) | symbol for codes of up to 9 bits (the host's look[]);
 *                       0x8000 | g for a prefix shared by longer codes; 0 = neither
 *   [512 + 128 g + b]   group g by the NEXT 7 bits: (length << 8) | symbol, lengths 10..16; 0 = no such code
 * Up to LUT_GROUPS prefixes get a group (the standard tables need 5-7); codes of further prefixes stay 0 in level one
 * and take the kernel's canonical-code walk. */
void build_lut(const struct huff &h, uint16_t *out)
{
    memset(out, 0, LUT_WORDS * sizeof(uint16_t));
    memcpy(out, h.look, 512 * sizeof(uint16_t));
    int groups = 0;
    for (int len = LOOK + 1; len <= 16; len++) {
        if (h.maxcode[len] < 0) continue;
        for (int code = h.mincode[len]; code <= h.maxcode[len]; code++) {
            const int idx = h.valptr[len] + code - h.mincode[len];
            if (idx < 0 || idx > 255 || code >= (1 << len)) return; /* a malformed DHT: leave the rest to the walk */
            const int prefix = code >> (len - LOOK);
            if (out[prefix] && !(out[prefix] & 0x8000)) return;       /* not prefix-free: leave it to the walk */
            if (!out[prefix]) {
                if (groups == LUT_GROUPS) continue;
                out[prefix] = (uint16_t)(0x8000 | groups++);
            }
            const int g = out[prefix] & 0xff;
            const int rest = (code << (16 - len)) & 127, cnt = 1 << (16 - len); /* the bits behind the prefix, left-aligned in 7 */
            for (int k = 0; k < cnt; k++) out[512 + 128 * g + rest + k] = (uint16_t)((len << 8) | h.vals[idx]);
        }
    }
}

template <typename F>
void parallel_for(int n, int n_threads, F f)
{
    if (n_threads > n) n_threads = n;
    if (n_threads <= 1) { for (int i = 0; i < n; i++) f(i); return; }
    std::vector<std::thread> pool;
    auto part = [&](int t) { for (int i = t; i < n; i += n_threads) f(i); };
    for (int t = 1; t < n_threads; t++) pool.emplace_back(part, t);
    part(0);
    for (auto &th : pool) th.join();
}
} // namespace

/* Test hook (host only, no device needed): stage_scan on caller memory.  dst must hold len + 8 n_seg + 64 bytes. */
extern "C" int ffhip_jpeg_stage_scan_test(uint8_t *dst, const uint8_t *src, size_t len, uint32_t *seg, uint32_t n_seg, size_t *clean_len)
{
    if (!dst || !src || !seg || !clean_len || n_seg == 0) return FFHIP_EINVAL;
    return (int)stage_scan(dst, src, src + len, seg, n_seg, clean_len);
}

/* what the calling thread's last ffhip_jpeg_entropy_batch_gpu call spent where (bench.py's configs.f1): microseconds of host time per phase,
 * and the Huffman kernel's own time by HIP events on the call's stream */
static thread_local double g_huff_times[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static thread_local hipEvent_t g_huff_ev[2] = {nullptr, nullptr};
extern "C" int ffhip_debug_huff_times(double out[8])
{
    if (!out) return FFHIP_EINVAL;
    for (int k = 0; k < 8; k++) out[k] = g_huff_times[k];
    return FFHIP_OK;
}

extern "C" int ffhip_jpeg_entropy_batch_gpu(const uint8_t *const *files, const size_t *lens, int n, int n_threads,
                                            const ffhip_jpeg_geom *geom, int16_t *d_coef_y, int16_t *d_coef_u, int16_t *d_coef_v,
                                            uint16_t *d_quant, int *status, void *stream)
{
    if (n < 0 || !geom || (n > 0 && (!files || !lens || !d_coef_y || !d_quant || !status))) return FFHIP_EINVAL;
    if (n == 0) return FFHIP_OK;
    if (geom->ncomp == 3 && (!d_coef_u || !d_coef_v)) return FFHIP_EINVAL;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (geom->mcu_cols <= 0 || geom->mcu_rows <= 0 || geom->h < 1 || geom->v < 1 || geom->h * geom->v > 4 ||
        (geom->ncomp != 1 && geom->ncomp != 3)) return FFHIP_EINVAL;
    const size_t mcus = (size_t)geom->mcu_cols * geom->mcu_rows;
    const bool times = FFHIP_ENV("FFHIP_HUFF_TIMES") != nullptr; /* host phases on stderr */
    const auto T0 = std::chrono::steady_clock::now();
    /* ---- host, pictures over threads: headers, tables, restart-interval starts ---- */
    std::vector<struct jpeg_hdr> hdr((size_t)n);
    std::vector<std::vector<uint32_t>> segs((size_t)n);
    parallel_for(n, n_threads, [&](int i) {
        struct jpeg_hdr &j = hdr[(size_t)i];
        status[i] = ffhip_jpeg_parse(files[i], lens[i], &j);
        if (status[i]) return;
        const int mc = (j.width + 8 * j.h[0] - 1) / (8 * j.h[0]), mr = (j.height + 8 * j.v[0] - 1) / (8 * j.v[0]);
        if (mc != geom->mcu_cols || mr != geom->mcu_rows || j.ncomp != geom->ncomp || j.h[0] != geom->h || j.v[0] != geom->v ||
            j.scan_len > 0x7fffffffu) {
            status[i] = FFHIP_EINVAL; /* another geometry */
            return;
        }
        if (!j.restart) j.restart = (int)mcus; /* no DRI: the whole scan is one interval = one lane (worth it for large batches only) */
        /* the interval starts are found while the bytes are staged (stage_scan) */
        segs[(size_t)i].assign((size_t)((mcus + j.restart - 1) / j.restart), 0u);
    });
    for (int i = 0; i < n; i++)
        if (status[i]) return status[i];
    const auto T1 = std::chrono::steady_clock::now();
    /* ---- layout of the one upload: scan bytes | tables | picture records | interval starts | work list | status ---- */
    std::vector<HuffImage> images((size_t)n);
    /* pictures of a batch mostly share their Huffman tables (an encoder's defaults): keep one copy of each distinct table */
    std::vector<const struct huff *> uniq;
    auto table_id = [&](const struct huff *t) -> uint32_t {
        for (size_t u = uniq.size(); u-- > 0;) /* newest first: the previous picture's are the likely match */
            if (uniq[u] == t || !memcmp(uniq[u], t, sizeof(struct huff))) return (uint32_t)u;
        uniq.push_back(t);
        return (uint32_t)(uniq.size() - 1);
    };
    size_t scan_total = 0, seg_total = 0;
    for (int i = 0; i < n; i++) {
        const struct jpeg_hdr &j = hdr[(size_t)i];
        HuffImage &im = images[(size_t)i];
        im.scan_off = (uint32_t)scan_total;
        im.scan_len = (uint32_t)j.scan_len;
        im.restart = (uint32_t)j.restart;
        im.mcus = (uint32_t)mcus;
        im.ncomp = (uint32_t)j.ncomp;
        im.seg_base = (uint32_t)seg_total;
        im.n_seg = (uint32_t)segs[(size_t)i].size();
        for (int c = 0; c < 3; c++) {
            im.nb[c] = c < j.ncomp ? (uint32_t)(j.h[c] * j.v[c]) : 0;
            im.tab_dc[c] = table_id(&j.dc[c < j.ncomp ? j.td[c] : j.td[0]]);
            im.tab_ac[c] = table_id(&j.ac[c < j.ncomp ? j.ta[c] : j.ta[0]]);
        }
        scan_total += (j.scan_len + 8 * (size_t)im.n_seg + 64 + 15) & ~(size_t)15; /* unstuffed, every interval aligned and padded, slack for the 16-byte stores */
        seg_total += im.n_seg;
        im.seg_base += (uint32_t)i; /* one more entry per picture: the end of its last interval */
        if (scan_total > 0x7fffffffu) return FFHIP_EINVAL;
    }
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    const size_t n_tabs = uniq.size();
    if (n_tabs > 4095) return FFHIP_EINVAL; /* table indices travel in 12 bits */
    const size_t o_tabs = scan_total + 16, o_l12 = (o_tabs + n_tabs * sizeof(struct huff) + 15) & ~(size_t)15;
    const size_t o_img = o_l12 + n_tabs * LUT_WORDS * 2;
    const size_t o_seg = (o_img + images.size() * sizeof(HuffImage) + 15) & ~(size_t)15, o_work = (o_seg + (seg_total + (size_t)n) * 4 + 15) & ~(size_t)15;
    const size_t o_status = (o_work + seg_total * 8 + 15) & ~(size_t)15, o_quant = (o_status + (size_t)n * 4 + 15) & ~(size_t)15;
    const size_t total = o_quant + (size_t)n * 512;
    /* pinned staging and device image are kept per stream: callers on different streams overlap completely */
    uint8_t *stage = ffhip_pinned_scratch(SCRATCH_HUFF, stream, total + 64);
    if (!stage) return FFHIP_ENOMEM;
    const auto T2 = std::chrono::steady_clock::now();
    hipStream_t st = (hipStream_t)stream;
    uint8_t *dev = (uint8_t *)ffhip_scratch(SCRATCH_HUFF, stream, (total + 3) / 4);
    if (!dev) return FFHIP_ENOMEM;
    /* the pictures are staged a quarter of the batch at a time and each quarter's bytes go up while the next is
     * being unstuffed: the upload (8 ms for 256 4K files) hides behind the staging (7 ms) instead of following it */
    /* the kernel stores the non-zero coefficients only: the planes are cleared first -- enqueued here, so that the
     * clears (4.8 GB for 256 4K pictures) run while the host is still staging */
    {
        const size_t yb = mcus * geom->h * geom->v * 64;
        FFHIP_CHECK(hipMemsetAsync(d_coef_y, 0, (size_t)n * yb * 2, st), FFHIP_EIO);
        if (geom->ncomp == 3) {
            FFHIP_CHECK(hipMemsetAsync(d_coef_u, 0, (size_t)n * mcus * 128, st), FFHIP_EIO);
            FFHIP_CHECK(hipMemsetAsync(d_coef_v, 0, (size_t)n * mcus * 128, st), FFHIP_EIO);
        }
    }
    const int n_parts = n >= 32 ? 4 : 1;
    for (int part = 0; part < n_parts; part++) {
    const int p_lo = (int)((long long)n * part / n_parts), p_hi = (int)((long long)n * (part + 1) / n_parts);
    parallel_for(p_hi - p_lo, n_threads, [&](int i_rel) {
        const int i = p_lo + i_rel;
        const struct jpeg_hdr &j = hdr[(size_t)i];
        const HuffImage &im = images[(size_t)i];
        /* the picture's bytes, unstuffed, every restart interval 4-byte aligned and followed by >= 4 zero bytes */
        uint8_t *dst = stage + im.scan_off;
        std::vector<uint32_t> &sgv = segs[(size_t)i];
        size_t off = 0;
        if (stage_scan(dst, j.scan, j.scan + j.scan_len, sgv.data(), im.n_seg, &off) != im.n_seg) status[i] = FFHIP_EINVAL;
        memset(dst + off, 0, 16);
        uint32_t *sg = (uint32_t *)(stage + o_seg) + im.seg_base;
        u32x2 *wk = (u32x2 *)(stage + o_work) + (im.seg_base - (uint32_t)i);
        for (uint32_t k = 0; k < im.n_seg; k++) {
            sg[k] = segs[(size_t)i][k];
            wk[k].x = (uint32_t)i;
            wk[k].y = k;
        }
        sg[im.n_seg] = (uint32_t)off;
        memcpy(stage + o_quant + (size_t)i * 512, j.quant, 512);
    });
    {
        const size_t b0 = images[(size_t)p_lo].scan_off, b1 = p_hi < n ? images[(size_t)p_hi].scan_off : scan_total;
        FFHIP_CHECK(hipMemcpyAsync(dev + b0, stage + b0, b1 - b0, hipMemcpyHostToDevice, st), FFHIP_EIO);
    }
    } /* parts */
    for (int i = 0; i < n; i++)
        if (status[i]) { /* a file whose restart markers do not add up: nothing is decoded */
            (void)hipStreamSynchronize(st);
            return status[i];
        }
    memset(stage + scan_total, 0, 16);
    const auto T3 = std::chrono::steady_clock::now();
    parallel_for((int)n_tabs, n_threads, [&](int u) {
        ((struct huff *)(stage + o_tabs))[u] = *uniq[(size_t)u];
        build_lut(*uniq[(size_t)u], (uint16_t *)(stage + o_l12) + (size_t)u * LUT_WORDS);
    });
    memcpy(stage + o_img, images.data(), images.size() * sizeof(HuffImage));
    memset(stage + o_status, 0, (size_t)n * 4);
    const auto T4 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    if (times) {
        fprintf(stderr, "huff staging: header parse %ld us, layout %ld us, unstuff + markers (uploads enqueued by quarters) %ld us, tables %ld us (%d files, %zu bytes, %d threads)\n", us(T0, T1), us(T1, T2), us(T2, T3), us(T3, T4), n, total, n_threads);
    }
    /* the rest of the image: scan padding, tables, picture records, interval lists, status, quantiser tables */
    FFHIP_CHECK(hipMemcpyAsync(dev + scan_total, stage + scan_total, total - scan_total, hipMemcpyHostToDevice, st), FFHIP_EIO);
    FFHIP_CHECK(hipMemcpyAsync(d_quant, dev + o_quant, (size_t)n * 512, hipMemcpyDeviceToDevice, st), FFHIP_EIO);
    HuffArgs a;
    a.scan = dev;
    a.tabs = (const struct huff *)(dev + o_tabs);
    a.lut = (const uint16_t *)(dev + o_l12);
    a.images = (const HuffImage *)(dev + o_img);
    a.seg = (const uint32_t *)(dev + o_seg);
    a.work = (const u32x2 *)(dev + o_work);
    a.plane[0] = d_coef_y; a.plane[1] = d_coef_u; a.plane[2] = d_coef_v;
    a.status = (int *)(dev + o_status);
    a.n_work = (uint32_t)seg_total;
    if (!g_huff_ev[0] && (hipEventCreate(&g_huff_ev[0]) != hipSuccess || hipEventCreate(&g_huff_ev[1]) != hipSuccess)) { (void)hipGetLastError(); g_huff_ev[0] = nullptr; }
    if (g_huff_ev[0]) (void)hipEventRecord(g_huff_ev[0], st);
    {   /* 128-byte rings (two workgroups of four waves per CU) while that holds the whole batch at once; 64-byte rings (three per CU, a refill every 8 symbols instead
         * of 16) beyond: 256 4K files of 135 intervals 9.9 ms against 10.4, 1 024 files 25.2 against 20.2 */
        int cus = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const unsigned wgs = (a.n_work + 255) / 256;
        const char *fr = FFHIP_ENV("FFHIP_HUFF_RING"); /* =16 / =32 forces either */
        if (fr ? atoi(fr) == 16 : wgs > 2u * (unsigned)cus) hipLaunchKernelGGL((k_jpeg_huff<16, 8>), dim3(wgs), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_jpeg_huff<32, 16>), dim3(wgs), dim3(256), 0, st, a);
    }
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    if (g_huff_ev[0]) (void)hipEventRecord(g_huff_ev[1], st);
    /* per-picture verdicts come back with the stream (tiny); the staging buffer is free again after this sync */
    FFHIP_CHECK(hipMemcpyAsync(stage + o_status, dev + o_status, (size_t)n * 4, hipMemcpyDeviceToHost, st), FFHIP_EIO);
    const auto T5 = std::chrono::steady_clock::now();
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
    const auto T6 = std::chrono::steady_clock::now();
    if (times)
        fprintf(stderr, "huff device: enqueue %ld us, wait for uploads + clears + kernel %ld us\n", us(T3, T5), us(T5, T6));
    {
        float kms = 0.0f;
        if (!g_huff_ev[0] || hipEventElapsedTime(&kms, g_huff_ev[0], g_huff_ev[1]) != hipSuccess) { (void)hipGetLastError(); kms = 0.0f; }
        g_huff_times[0] = (double)us(T0, T1); g_huff_times[1] = (double)us(T1, T2); g_huff_times[2] = (double)us(T2, T3); g_huff_times[3] = (double)us(T3, T4);
        g_huff_times[4] = (double)us(T4, T5); g_huff_times[5] = (double)us(T5, T6); g_huff_times[6] = (double)kms * 1e3; g_huff_times[7] = (double)us(T0, T6);
    }
    memcpy(status, stage + o_status, (size_t)n * 4);
    for (int i = 0; i < n; i++)
        if (status[i]) return status[i];
    return FFHIP_OK;
}
