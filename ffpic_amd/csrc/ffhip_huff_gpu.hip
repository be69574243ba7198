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

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>
#include <memory>
#include <new>
#include <type_traits>

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
#define LUT_NO_CODE 0x5000u /* (16 << 8) | bit 14: no code starts with these bits */
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
            int sym = (int)(e & 0xff), len = (int)((e >> 8) & 31u);
            bad |= (e & 0x4000u) != 0; /* no such code (LUT_NO_CODE) */
            if (!e) { /* a table with more long-code groups than the LUT holds, or a malformed one */
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
#define SCRATCH_HUFF_SYNC 30

namespace {
/* One pass over a picture's entropy-coded segment: the bytes without their stuffing (FF 00 -> FF) into dst, every
 * restart interval 4-byte aligned and followed by at least 4 zero bytes, seg[k] = offset of interval k in that clean
 * stream.  RSTn markers separate the intervals; any other marker (or the n_seg-th RSTn) ends the scan.  Returns the
 * number of intervals found; *clean_len = bytes written.  16 bytes at a time while no 0xFF is in sight (one in 256
 * bytes is, in entropy-coded data): the first form, memchr + memcpy per run and a separate marker scan before it,
 * spent its time in call overhead -- 7 ms of staging per 256 4K files on 16 threads, against 12 ms of kernel.
 * May store up to 15 bytes past the clean length: the caller reserves the slack. */
uint32_t stage_scan(uint8_t *dst, const uint8_t *src, const uint8_t *end, uint32_t *seg, uint32_t n_seg, size_t *clean_len, uint32_t *raw = nullptr)
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
            if (raw) raw[k] = (uint32_t)len; /* the interval's own bytes, without the padding */
            seg[++k] = (uint32_t)(d - dst);
            src += 2;
        }
    }
    {
        const size_t len = (size_t)(d - dst) - seg[k], padded = ((len + 3) & ~(size_t)3) + 4;
        memset(d, 0, padded - len);
        d = dst + seg[k] + padded;
        if (raw) raw[k] = (uint32_t)len;
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
    bool complete = true;
    for (int len = LOOK + 1; len <= 16; len++) {
        if (h.maxcode[len] < 0) continue;
        for (int code = h.mincode[len]; code <= h.maxcode[len]; code++) {
            const int idx = h.valptr[len] + code - h.mincode[len];
            if (idx < 0 || idx > 255 || code >= (1 << len)) return; /* a malformed DHT: leave the rest to the walk */
            const int prefix = code >> (len - LOOK);
            if (out[prefix] && !(out[prefix] & 0x8000)) return;       /* not prefix-free: leave it to the walk */
            if (!out[prefix]) {
                if (groups == LUT_GROUPS) { complete = false; continue; }
                out[prefix] = (uint16_t)(0x8000 | groups++);
            }
            const int g = out[prefix] & 0xff;
            const int rest = (code << (16 - len)) & 127, cnt = 1 << (16 - len); /* the bits behind the prefix, left-aligned in 7 */
            for (int k = 0; k < cnt; k++) out[512 + 128 * g + rest + k] = (uint16_t)((len << 8) | h.vals[idx]);
        }
    }
    /* every code of the table is in: what is still 0 is no code at all.  Say so in the entry (length 16, bit 14), so that a lane on damaged data -- or one
     * of the subsequence decoder's, started at a wrong bit -- takes one look-up to find out instead of the canonical-code walk through global memory */
    if (complete)
        for (int i = 0; i < 512 + 128 * groups; i++)
            if (!out[i]) out[i] = LUT_NO_CODE;
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

/* The same with the intervals' own lengths (without padding) in raw[0 .. n_seg): what the subsequence decoder cuts into lanes. */
extern "C" int ffhip_jpeg_stage_scan_raw_test(uint8_t *dst, const uint8_t *src, size_t len, uint32_t *seg, uint32_t n_seg, size_t *clean_len, uint32_t *raw)
{
    if (!dst || !src || !seg || !clean_len || !raw || n_seg == 0) return FFHIP_EINVAL;
    return (int)stage_scan(dst, src, src + len, seg, n_seg, clean_len, raw);
}
/* Test hook (host only): the two-level look-up table the device kernels use for Huffman table `which` (0..3 DC, 4..7 AC) of a file, LUT_WORDS = 1536
 * uint16 (build_lut above).  FFHIP_EINVAL if the file does not parse or has no such table. */
extern "C" int ffhip_jpeg_lut_test(const uint8_t *file, size_t len, int which, uint16_t *out)
{
    if (!file || !out || which < 0 || which > 7) return FFHIP_EINVAL;
    std::unique_ptr<struct jpeg_hdr> j(new (std::nothrow) struct jpeg_hdr);
    if (!j) return FFHIP_ENOMEM;
    if (ffhip_jpeg_parse(file, len, j.get())) return FFHIP_EINVAL;
    const struct huff &h = which < 4 ? j->dc[which] : j->ac[which - 4];
    if (!h.present) return FFHIP_EINVAL;
    build_lut(h, out);
    return FFHIP_OK;
}

/* what the calling thread's last ffhip_jpeg_entropy_batch_gpu call spent where (bench.py's configs.f1): microseconds of host time per phase,
 * and the Huffman kernel's own time by HIP events on the call's stream */
static thread_local double g_huff_times[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SYNC_PARTS FFHIP_HUFF_PARTS
thread_local FfhipHuffThen g_ffhip_huff_then = {0, nullptr, 0, 0};
/* (the upload stream, the second kernel stream and the events of a call are the thread's and the device's: ffhip_huff_streams_get) */
/* bits of a subsequence, unless FFHIP_JPEG_SYNC_BITS sets them: by the bits an MCU takes (huff_sync_enqueue has the measurements) */
static uint32_t sync_sub_bits(unsigned long long bits, unsigned long long mcus)
{
    const char *b = FFHIP_ENV("FFHIP_JPEG_SYNC_BITS");
    const long sb = b ? atol(b) : bits <= 1024 * mcus ? 2048 : bits <= 2048 * mcus ? 4096 : 8192;
    return (uint32_t)(sb < 128 ? 128 : sb > 65536 ? 65536 : sb);
}
extern "C" int ffhip_debug_huff_times(double out[8])
{
    if (!out) return FFHIP_EINVAL;
    for (int k = 0; k < 8; k++) out[k] = g_huff_times[k];
    return FFHIP_OK;
}

/* files without restart markers: the self-synchronising subsequence decoder at the end of this file */
struct SyncSeg { /* a restart interval of a picture of the part (a file without restart markers: its whole scan) */
    uint32_t pic;                 /* picture of the part */
    uint32_t scan_off, clean_len; /* in the call's device image; staged bytes with the zero padding */
    uint32_t raw_len;             /* the interval's own bytes */
    uint32_t mcu0, mcus;          /* first MCU (counted over the part's pictures) and MCUs of the interval */
};
struct SyncJob {
    uint8_t *dev;                 /* the call's device image: scan bytes, tables, look-up tables, status */
    size_t o_tabs, o_l12, o_status;
    int n, part;                  /* pictures of the part */
    void *stream;                 /* the stream the part's kernels are on */
    const HuffImage *images;      /* tables and block counts as for the kernel above */
    const SyncSeg *segs;
    size_t n_segs;
    int16_t *plane[3];
    uint32_t rounds_used;         /* out: synchronisation rounds that changed something (diagnostics) */
    uint32_t n_tasks;             /* out */
    int reran;                    /* out: the rounds launched at first did not do; the passes ran from huff_sync_finish */
    uint32_t rounds;              /* out: list rounds per batch of launches */
    uint32_t sub_bits;            /* out: bits of a subsequence */
};
static int huff_sync_enqueue(SyncJob &job, void *stream, uint32_t **h_cnt);
static int huff_sync_finish(SyncJob &job, void *stream, uint32_t *h_cnt, int *status);

/* the header records of a batch, kept by the thread between calls (84 MB of fresh pages for 4 096 files, and their return, are milliseconds);
 * ffhip_release_caches lets the calling thread's go */
static thread_local std::unique_ptr<struct jpeg_hdr[]> hdr_keep;
static thread_local size_t hdr_cap = 0;
extern "C" void ffhip_huff_release_thread(void)
{
    hdr_keep.reset();
    hdr_cap = 0;
}

extern "C" int ffhip_jpeg_entropy_batch_gpu(const uint8_t *const *files, const size_t *lens, int n, int n_threads,
                                            const ffhip_jpeg_geom *geom, int16_t *d_coef_y, int16_t *d_coef_u, int16_t *d_coef_v,
                                            uint16_t *d_quant, int *status, void *stream)
{
    const FfhipHuffThen then = g_ffhip_huff_then;
    g_ffhip_huff_then.on = 0;
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
    /* (not a std::vector: that would zero 20 KB a file on this thread before the parsing threads start -- 84 MB and 10 ms for 4 096 thumbnails -- and
     * ffhip_jpeg_parse clears its record itself) */
    if ((size_t)n > hdr_cap) {
        hdr_keep.reset(new (std::nothrow) struct jpeg_hdr[(size_t)n + (size_t)n / 4]);
        hdr_cap = hdr_keep ? (size_t)n + (size_t)n / 4 : 0;
        if (!hdr_keep) return FFHIP_ENOMEM;
    }
    struct jpeg_hdr *const hdr = hdr_keep.get();
    std::vector<std::vector<uint32_t>> segs((size_t)n);
    std::vector<std::vector<uint32_t>> raws((size_t)n); /* per picture: its intervals' own lengths */
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
        if (!j.restart) j.restart = (int)mcus; /* no DRI: the whole scan is one interval */
        /* the interval starts are found while the bytes are staged (stage_scan) */
        segs[(size_t)i].assign((size_t)((mcus + j.restart - 1) / j.restart) + 1, 0u); /* (one more: the end of the last) */
        raws[(size_t)i].assign(segs[(size_t)i].size(), 0u);
    });
    for (int i = 0; i < n; i++)
        if (status[i]) return status[i];
    /* the subsequence decoder: a lane per 2048 bits of a restart interval (of the whole scan, in a file without restart markers), brought into step with
     * each other over rounds.  FFHIP_JPEG_SYNC=0: the kernel above, a lane per restart interval -- a file without markers is ONE lane's then */
    const char *sy = FFHIP_ENV("FFHIP_JPEG_SYNC");
    bool use_sync = !(sy && sy[0] == '0');
    /* the subsequences' length is worked out ONCE, from the batch's scan bytes: the choice of kernel below and every part's passes go by the same figure */
    unsigned long long batch_bits = 0;
    for (int i = 0; i < n; i++) batch_bits += 8ull * hdr[(size_t)i].scan_len;
    const uint32_t sub_bits = sync_sub_bits(batch_bits, (unsigned long long)n * mcus);
    if (use_sync && !(sy && sy[0] == '1')) {
        /* restart intervals of a subsequence or two (a DRI of one or a few MCUs: 32 400 intervals in a 4K picture) are lanes enough as they are, every
         * one starting from the truth: three passes, a 60-byte record per interval and rounds that have nothing to settle are the wrong tool; the kernel
         * above takes such batches (FFHIP_JPEG_SYNC=1 keeps the subsequence decoder on them) */
        unsigned long long n_int = 0;
        for (int i = 0; i < n; i++) n_int += segs[(size_t)i].size() - 1;
        if (batch_bits <= 2ull * sub_bits * n_int) use_sync = false;
    }
    const auto T1 = std::chrono::steady_clock::now();
    /* ---- layout of the one upload: scan bytes | tables | picture records | interval starts | work list | status ---- */
    std::vector<HuffImage> images((size_t)n);
    /* pictures of a batch mostly share their Huffman tables (an encoder's defaults): keep one copy of each distinct table */
    std::vector<const struct huff *> uniq;
    auto table_id = [&](const struct huff *t) -> uint32_t {
        for (size_t u = uniq.size(); u-- > 0;) /* newest first: the previous picture's are the likely match */
            if (uniq[u] == t || !memcmp(&uniq[u]->maxcode, &t->maxcode, offsetof(struct huff, fast) - offsetof(struct huff, maxcode))) return (uint32_t)u; /* (look[] and fast[] follow from the rest) */
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
        im.n_seg = (uint32_t)segs[(size_t)i].size() - 1;
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
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    /* tables, picture records, status, quantiser tables: behind the scan bytes in the image.  The files with markers send them last (the interval lists are
     * found while staging); the subsequence decoder first, because it starts on a part of the batch as soon as that part's bytes are up */
    long tables_us = 0;
    auto tail_up = [&]() -> int {
        const auto Ta = std::chrono::steady_clock::now();
        memset(stage + scan_total, 0, 16);
        parallel_for((int)n_tabs, n_threads, [&](int u) {
            ((struct huff *)(stage + o_tabs))[u] = *uniq[(size_t)u];
            build_lut(*uniq[(size_t)u], (uint16_t *)(stage + o_l12) + (size_t)u * LUT_WORDS);
        });
        memcpy(stage + o_img, images.data(), images.size() * sizeof(HuffImage));
        memset(stage + o_status, 0, (size_t)n * 4);
        tables_us = us(Ta, std::chrono::steady_clock::now());
        FFHIP_CHECK(hipMemcpyAsync(dev + scan_total, stage + scan_total, total - scan_total, hipMemcpyHostToDevice, st), FFHIP_EIO);
        FFHIP_CHECK(hipMemcpyAsync(d_quant, dev + o_quant, (size_t)n * 512, hipMemcpyDeviceToDevice, st), FFHIP_EIO);
        return FFHIP_OK;
    };
    FfhipHuffStreams hs;
    if (ffhip_huff_streams_get(&hs) != FFHIP_OK) return FFHIP_EIO; /* (nothing of this call is enqueued yet but the plane clears, which touch no buffer of the library's) */
    const hipStream_t g_huff_up = (hipStream_t)hs.up, g_huff_c2 = (hipStream_t)hs.c2;
    hipEvent_t *const g_huff_part_ev = (hipEvent_t *)hs.part_ev;
    const hipEvent_t g_huff_fork = (hipEvent_t)hs.fork, g_huff_join = (hipEvent_t)hs.join;
    const hipEvent_t g_huff_ev[2] = {(hipEvent_t)hs.time_ev[0], (hipEvent_t)hs.time_ev[1]};
    /* every failure behind this point leaves through `fail`: uploads, part kernels or a `then` reconstruction into the caller's pixels may be in flight,
     * and the pinned stage, the device image and the parts' scratch are refilled by the thread's next call */
    auto fail = [&](int code) -> int {
        (void)hipStreamSynchronize(g_huff_up);
        (void)hipStreamSynchronize(g_huff_c2);
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        return code;
    };
#define HUFF_CHECK(call) do { if ((call) != hipSuccess) return fail(FFHIP_EIO); } while (0)
    int n_parts = n >= 32 ? 4 : 1;
    if (use_sync) {
        /* parts of about 140 MB of scan bytes, a hundred 4K files -- a round over fewer subsequences fills the chip badly, and what is left to wait for
         * behind the last upload are the last part's kernels: 256 4K files 13.7-14.0 ms in three parts, 14.0-14.5 in four, 14.2 in two, 14.4-14.5 in six (and 14.7-14.9 in four of
         * 1/8, 3/8, 3/8, 1/8 of the files, which looked good on paper) */
        const int by_bytes = (int)((scan_total + (70u << 20)) / (140u << 20)); /* (by bytes, not by files: 4 096 thumbnails are a hundred megabytes) */
        n_parts = n < 32 || scan_total < (16u << 20) ? 1 : by_bytes < 2 ? 2 : by_bytes > SYNC_PARTS ? SYNC_PARTS : by_bytes;
        const char *e = FFHIP_ENV("FFHIP_JPEG_SYNC_PARTS"); /* parts of the batch that are staged, sent and decoded one behind the other (1..8) */
        if (e && atoi(e) >= 1) n_parts = atoi(e) > SYNC_PARTS ? SYNC_PARTS : atoi(e);
        if (n_parts > n) n_parts = n;
    }
    SyncJob jobs[SYNC_PARTS];
    bool two_streams = false;
    std::vector<SyncSeg> part_segs[SYNC_PARTS];
    uint32_t *h_cnt[SYNC_PARTS];
    if (use_sync) {
        /* the parts' bytes go up on a stream of their own, the copy engine's, while the rounds of the parts before run on the caller's: 256 4K files are
         * 8 ms of PCIe and 11 ms of kernels */
        for (int i = 0; i < n; i++) memcpy(stage + o_quant + (size_t)i * 512, hdr[(size_t)i].quant, 512);
        const int rc = tail_up();
        if (rc) return fail(rc);
        (void)hipEventRecord(g_huff_ev[0], st);
        /* the parts' kernels alternate between the caller's stream and one of the library's: the list rounds of a part are a handful of sparse kernels
         * that each take as long as one lane takes for its subsequence, and run under the next part's full rounds instead of in front of them
         * (FFHIP_JPEG_SYNC_STREAMS=1: all on the caller's) */
        const char *ss = FFHIP_ENV("FFHIP_JPEG_SYNC_STREAMS");
        two_streams = n_parts > 1 && !(ss && ss[0] == '1');
        if (two_streams) HUFF_CHECK(hipEventRecord(g_huff_fork, st)); /* behind the plane clears, the tables and the quantiser copy */
        if (two_streams) HUFF_CHECK(hipStreamWaitEvent(g_huff_c2, g_huff_fork, 0));
    }
    auto part_lo = [&](int part) -> int { return (int)((long long)n * part / n_parts); };
    for (int part = 0; part < n_parts; part++) {
    const int p_lo = part_lo(part), p_hi = part_lo(part + 1);
    parallel_for(p_hi - p_lo, n_threads, [&](int i_rel) {
        const int i = p_lo + i_rel;
        const struct jpeg_hdr &j = hdr[(size_t)i];
        const HuffImage &im = images[(size_t)i];
        /* the picture's bytes, unstuffed, every restart interval 4-byte aligned and followed by >= 4 zero bytes */
        uint8_t *dst = stage + im.scan_off;
        std::vector<uint32_t> &sgv = segs[(size_t)i];
        size_t off = 0;
        if (stage_scan(dst, j.scan, j.scan + j.scan_len, sgv.data(), im.n_seg, &off, raws[(size_t)i].data()) != im.n_seg) status[i] = FFHIP_EINVAL;
        memset(dst + off, 0, 16);
        sgv[im.n_seg] = (uint32_t)off;
        if (use_sync) return; /* (the interval lists stay on the host; the tail of the image is on its way already) */
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
        if (!use_sync) {
            HUFF_CHECK(hipMemcpyAsync(dev + b0, stage + b0, b1 - b0, hipMemcpyHostToDevice, st));
        } else {
            SyncJob &job = jobs[part];
            int rc = FFHIP_OK;
            void *const pstream = two_streams && (part & 1) ? (void *)g_huff_c2 : stream; /* this part's kernels */
            job.stream = pstream;
            if (hipMemcpyAsync(dev + b0, stage + b0, b1 - b0, hipMemcpyHostToDevice, g_huff_up) != hipSuccess || hipEventRecord(g_huff_part_ev[part], g_huff_up) != hipSuccess ||
                hipStreamWaitEvent((hipStream_t)pstream, g_huff_part_ev[part], 0) != hipSuccess) rc = FFHIP_EIO;
            if (!rc) {
                job.sub_bits = sub_bits;
                job.dev = dev; job.o_tabs = o_tabs; job.o_l12 = o_l12; job.o_status = o_status + (size_t)p_lo * 4; job.n = p_hi - p_lo; job.part = part;
                job.images = images.data() + p_lo;
                part_segs[part].clear();
                for (int i = p_lo; i < p_hi && !rc; i++) {
                    const HuffImage &im = images[(size_t)i];
                    const std::vector<uint32_t> &sgv = segs[(size_t)i];
                    if (status[i]) { rc = status[i]; break; } /* a file whose restart markers do not add up */
                    for (uint32_t k = 0; k < im.n_seg; k++) {
                        SyncSeg sg;
                        sg.pic = (uint32_t)(i - p_lo);
                        sg.scan_off = im.scan_off + sgv[k];
                        sg.clean_len = sgv[k + 1] - sgv[k];
                        sg.raw_len = raws[(size_t)i][k];
                        sg.mcu0 = (uint32_t)((size_t)(i - p_lo) * mcus + (size_t)k * im.restart);
                        const size_t left = mcus - (size_t)k * im.restart;
                        sg.mcus = (uint32_t)(left < im.restart ? left : im.restart);
                        part_segs[part].push_back(sg);
                    }
                }
                job.segs = part_segs[part].data();
                job.n_segs = part_segs[part].size();
                if (rc) return fail(rc);
                job.plane[0] = d_coef_y + (size_t)p_lo * mcus * images[0].nb[0] * 64;
                job.plane[1] = d_coef_u ? d_coef_u + (size_t)p_lo * mcus * images[0].nb[1] * 64 : nullptr;
                job.plane[2] = d_coef_v ? d_coef_v + (size_t)p_lo * mcus * images[0].nb[2] * 64 : nullptr;
                rc = huff_sync_enqueue(job, pstream, &h_cnt[part]);
                if (!rc && then.on) /* the part's pictures: coefficients -> BGRA while the next part's bytes come up */
                    rc = ffhip_jpeg_recon_batch(geom, p_hi - p_lo, job.plane[0], job.plane[1], job.plane[2], d_quant + (size_t)p_lo * 256, 256,
                                                then.bgra + (int64_t)p_lo * then.image_stride, then.pitch, then.image_stride, nullptr, 0, pstream);
            }
            if (rc) return fail(rc); /* nothing of this call may be in flight when its buffers are handed back */
        }
    }
    } /* parts */
    if (two_streams) { /* the caller's stream is behind everything again */
        HUFF_CHECK(hipEventRecord(g_huff_join, g_huff_c2));
        HUFF_CHECK(hipStreamWaitEvent(st, g_huff_join, 0));
    }
    for (int i = 0; i < n; i++)
        if (status[i]) return fail(status[i]); /* a file whose restart markers do not add up: nothing is decoded */
    const auto T3 = std::chrono::steady_clock::now();
    if (!use_sync) {
        const int rc = tail_up();
        if (rc) return fail(rc);
    }
    const auto T4 = std::chrono::steady_clock::now();
    if (times) {
        fprintf(stderr, "huff staging: header parse %ld us, layout %ld us, unstuff + markers (uploads enqueued by parts) %ld us, tables %ld us (%d files, %zu bytes, %d threads)\n", us(T0, T1), us(T1, T2), us(T2, T3) - (use_sync ? tables_us : 0), tables_us, n, total, n_threads);
    }
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
    if (!use_sync) (void)hipEventRecord(g_huff_ev[0], st);
    if (!use_sync)
    {   /* 128-byte rings (two workgroups of four waves per CU) while that holds the whole batch at once; 64-byte rings (three per CU, a refill every 8 symbols instead
         * of 16) beyond: 256 4K files of 135 intervals 9.9 ms against 10.4, 1 024 files 25.2 against 20.2 */
        int cus = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const unsigned wgs = (a.n_work + 255) / 256;
        const char *fr = FFHIP_ENV("FFHIP_HUFF_RING"); /* =16 / =32 forces either */
        if (fr ? atoi(fr) == 16 : wgs > 2u * (unsigned)cus) hipLaunchKernelGGL((k_jpeg_huff<16, 8>), dim3(wgs), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_jpeg_huff<32, 16>), dim3(wgs), dim3(256), 0, st, a);
    }
    HUFF_CHECK(hipGetLastError());
    (void)hipEventRecord(g_huff_ev[1], st);
    if (!use_sync && then.on) {
        const int rc = ffhip_jpeg_recon_batch(geom, n, d_coef_y, d_coef_u, d_coef_v, d_quant, 256, then.bgra, then.pitch, then.image_stride, nullptr, 0, stream);
        if (rc) return fail(rc);
    }
    /* per-picture verdicts come back with the stream (tiny); the staging buffer is free again after this sync */
    HUFF_CHECK(hipMemcpyAsync(stage + o_status, dev + o_status, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    const auto T5 = std::chrono::steady_clock::now();
    HUFF_CHECK(hipStreamSynchronize(st));
    const auto T6 = std::chrono::steady_clock::now();
    if (times)
        fprintf(stderr, "huff device: enqueue %ld us, wait for uploads + clears + kernel %ld us\n", us(T3, T5), us(T5, T6));
    {
        float kms = 0.0f;
        if (hipEventElapsedTime(&kms, g_huff_ev[0], g_huff_ev[1]) != hipSuccess) { (void)hipGetLastError(); kms = 0.0f; }
        g_huff_times[0] = (double)us(T0, T1); g_huff_times[1] = (double)us(T1, T2); g_huff_times[2] = (double)(us(T2, T3) - (use_sync ? tables_us : 0)); g_huff_times[3] = (double)tables_us;
        g_huff_times[4] = (double)us(T4, T5); g_huff_times[5] = (double)us(T5, T6); g_huff_times[6] = (double)kms * 1e3; g_huff_times[7] = (double)us(T0, T6);
    }
    memcpy(status, stage + o_status, (size_t)n * 4);
    if (use_sync)
        for (int part = 0; part < n_parts; part++) {
            SyncJob &job = jobs[part];
            const int p_lo = (int)((job.o_status - o_status) / 4);
            int rc = huff_sync_finish(job, job.stream, h_cnt[part], status + p_lo);
            if (!rc && job.reran && then.on) { /* the part's passes ran only now: so must its reconstruction (and the caller's stream be behind it) */
                rc = ffhip_jpeg_recon_batch(geom, job.n, job.plane[0], job.plane[1], job.plane[2], d_quant + (size_t)p_lo * 256, 256,
                                            then.bgra + (int64_t)p_lo * then.image_stride, then.pitch, then.image_stride, nullptr, 0, job.stream);
                if (!rc && job.stream != stream && hipStreamSynchronize((hipStream_t)job.stream) != hipSuccess) rc = FFHIP_EIO;
            }
            if (rc) return fail(rc);
            if (times) fprintf(stderr, "huff sync, part %d: %u subsequences of %u bits, %u rounds\n", part, job.n_tasks, job.sub_bits, job.rounds_used);
        }
    for (int i = 0; i < n; i++)
        if (status[i]) return status[i];
    return FFHIP_OK;
#undef HUFF_CHECK
}

/* =====================================================================================================================
 * Files WITHOUT restart markers on the device (round 5): the scan cut into subsequences of `sub_bits` bits, a lane each.
 *
 * A baseline scan is one bit-serial stream per picture, and most files carry no DRI segment: to the kernel above such a
 * file is one interval = one lane (1.3 s of it for a 4K picture), so until now these files went to the host threads (6-8
 * Gpixel/s for 256 4K files on sixteen threads, the GPU idle).  Huffman-coded data SELF-SYNCHRONISES: a decoder started
 * at a wrong bit position falls into step with the true decode after a few symbols, and one in a wrong block state --
 * coefficient index, block slot of the MCU -- after a few blocks, MCUs at worst, because luma and chroma tables differ
 * (the observation parallel JPEG decoders are built on: Klein & Wiseman 2003; Weissenberger & Schmidt, "Accelerating JPEG
 * decompression on GPUs", 2021).  So:
 *   round 0     lane t decodes subsequence t from ITS first bit with a guessed state (first block of an MCU, k = 0) and
 *               records where and in which state it crossed into t + 1, how many blocks it completed and the sum of
 *               the DC differences it decoded, per component;
 *   round 1     every lane again, now from what lane t - 1 recorded;
 *   round r     a list of the lanes whose entry is no longer what they used (the lane in front changed its exit), and
 *               those lanes again, packed into waves: a picture's first lane always starts from the truth, so by
 *               induction the fixed point IS the sequential decode; it is reached when a list comes out empty (a
 *               handful of rounds; never more than the longest picture has subsequences);
 *   scan        per picture the exclusive prefix sums of blocks and DC sums over its lanes: every lane's first block
 *               and its DC predictors (jpg.c:255-415's running sums) there;
 *   write       every lane decodes its subsequence once more, now storing coefficients.
 * The host does what it does for the files with markers: headers, tables, unstuffing into pinned memory.
 * ===================================================================================================================== */
#define SYNC_ROUNDS_MAX 32 /* list rounds per batch of launches; FFHIP_JPEG_SYNC_ROUNDS (1..32), 10 unless set */

struct SyncImage {
    uint32_t scan_off;   /* byte offset of the picture's unstuffed scan inside `scan` (16-byte aligned) */
    uint32_t clean_len;  /* bytes staged: the data and its zero padding                                 */
    uint32_t data_bits;  /* bits of entropy-coded data (without the padding)                            */
    uint32_t mcus, ncomp, nbt; /* MCUs of the interval; nbt: blocks per MCU over all components         */
    uint32_t nb[3], tab_dc[3], tab_ac[3];
    uint32_t pic, mcu0;  /* picture of the part (status), first MCU counted over the part's pictures    */
};
struct SyncArgs {
    const uint8_t *scan;
    const struct huff *tabs;
    const uint16_t *lut;
    const SyncImage *images;
    const uint32_t *sub_base;    /* [n_images + 1]: first task of a picture */
    unsigned long long *exit_;   /* per task: bit position | (slot | k << 8) << 32 where the NEXT task starts */
    unsigned long long *used;    /* per task: the entry its exit was worked out from                           */
    u32x4 *sums;                 /* per task: blocks completed, DC differences summed per component; after the scan their exclusive prefix sums */
    uint32_t *list;              /* the tasks of a list round */
    uint32_t *seg_of;            /* per task: its interval (filled by k_huff_sync_segs; looked up, not searched for, by every pass) */
    uint32_t *cnt;               /* [last + 1]: tasks in round r's list; cnt[last] == 0: the fixed point is reached */
    uint32_t *end_pos;           /* per picture: bit position behind its last block (write pass)               */
    int16_t *plane[3];
    int *status;
    uint32_t n_tasks, n_images, round, last, sub_bits;
};

__device__ __forceinline__ uint32_t sync_picture_of(const SyncArgs &a, uint32_t t)
{ /* the interval ("picture" of these kernels) task t belongs to */
    return a.seg_of[t];
}
/* one workgroup per interval writes the interval's index over its tasks: every pass starts with one load per lane instead of a binary search of 17
 * loads that depend on each other (8 us a wave; the list kernels were nothing but that search) */
__global__ __launch_bounds__(256) void k_huff_sync_segs(SyncArgs a)
{
    const uint32_t p = blockIdx.x, t1 = a.sub_base[p + 1];
    for (uint32_t t = a.sub_base[p] + threadIdx.x; t < t1; t += 256) a.seg_of[t] = p;
}

/* the tasks whose entry is not what they used: round `a.round`'s list.  The append is aggregated per workgroup of sixteen waves -- one atomic on the
 * list's counter per 1 024 tasks: with one per wave the kernel was 0.29 ms of atomics on one address for 1.6 M tasks.  The order does not matter. */
#define LIST_THREADS 1024
__global__ __launch_bounds__(LIST_THREADS) void k_huff_sync_list(SyncArgs a)
{
    __shared__ uint32_t wcount[LIST_THREADS / 64], wbase[LIST_THREADS / 64];
    if (a.round >= 3 && a.cnt[a.round - 1] == 0) return; /* the list before was empty: so is this one */
    const uint32_t t = blockIdx.x * LIST_THREADS + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool stale = false;
    if (t > 0 && t < a.n_tasks) stale = a.seg_of[t - 1] == a.seg_of[t] && a.exit_[t - 1] != a.used[t]; /* (an interval's first task starts from the truth) */
    const unsigned long long m = __builtin_amdgcn_ballot_w64(stale);
    if (lane == 0) wcount[wv] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < LIST_THREADS / 64; w++) { wbase[w] = total; total += wcount[w]; }
        const uint32_t base = total ? atomicAdd(&a.cnt[a.round], total) : 0u;
        for (int w = 0; w < LIST_THREADS / 64; w++) wbase[w] += base;
    }
    __syncthreads();
    if (stale) a.list[wbase[wv] + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = t;
}

enum { SPAN_ALL = 0, SPAN_LIST = 1, SPAN_WRITE = 2 };
#define SPAN_THREADS 512
#define WRITE_THREADS 256
/* What bounds this kernel is the SIMD's vector ALU -- a wave64 instruction takes four cycles, and a symbol took some 80 of them in the first form (64-bit
 * shifts, predictors and block addresses looked up and selected at every block change, the compiler's copies around a tangled refill loop) -- and, below
 * that, LDS for the waves per SIMD.  So:
 *   - the bit reader is two dwords and a shift count: the next 32 bits are ONE v_alignbit, the magnitude bits one v_bfe; a symbol takes at most 31 bits;
 *   - EIGHT waves to a workgroup share the six look-up tables (18 KB), and nothing else is in LDS but each lane's 64-byte ring: three workgroups per CU,
 *     six waves per SIMD; the MCU's block slots are packed words (one geometry per call), the predictors three registers that ROTATE when the component
 *     changes (the components of an interleaved scan follow each other in a cycle);
 *   - the ring is topped up one refill point AHEAD -- a chunk is loaded at one point and put into the ring at the next, so the wave does not wait for
 *     the load; a point every four symbols and a chunk of four dwords each keep a full ring ahead of any stream (31 bits a symbol at most);
 *   - a look-up table that holds every code of its Huffman table answers "no such code" itself (build_lut), so lanes decoding from a wrong entry do
 *     not walk the canonical code through global memory. */
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_huff_span(SyncArgs a)
{
    constexpr bool WRITE = MODE == SPAN_WRITE;
    constexpr int RING = 16, REFILL = 4, WAVES = THREADS / 64;
    constexpr uint32_t LUT_BYTES = LUT_WORDS * 2;
    __shared__ uint8_t zz[64];
    __shared__ uint16_t lt[6][LUT_WORDS];
    __shared__ uint32_t ring[WAVES][RING][64];
    /* the write pass: a lane's block is put together in LDS -- its first four rows, the 64 bytes nearly all coefficients of a photograph land in -- and
     * goes out row by row, 16 bytes a store, when the block is complete: a store of TWO bytes costs the L2 what one of sixteen does, and with nine
     * coefficients a block that was 6 of the pass's 7.4 ms */
    __shared__ u32x4 stage[WRITE ? 4 : 1][WRITE ? THREADS : 1]; /* [row][thread]: the row's eight coefficients, one 16-byte LDS access to take out and clear */
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t i0 = blockIdx.x * THREADS;
    uint32_t n_here = a.n_tasks;
    if (MODE == SPAN_LIST) {
        n_here = a.cnt[a.round];
        if (i0 >= n_here) return;            /* (the whole workgroup) */
    }
    if (WRITE && a.cnt[a.last]) return;      /* the rounds launched did not reach the fixed point: the host adds rounds and launches this again */
    if (threadIdx.x < 64) zz[threadIdx.x] = kZigzag[threadIdx.x];
    const uint32_t t_first = MODE == SPAN_LIST ? a.list[i0] : i0;
    const uint32_t p0 = sync_picture_of(a, t_first);
    const SyncImage im0 = a.images[p0];
    for (int k2 = 0; k2 < 3; k2++) {
        const u32x4 *sd = (const u32x4 *)(a.lut + (size_t)im0.tab_dc[k2] * LUT_WORDS), *sa = (const u32x4 *)(a.lut + (size_t)im0.tab_ac[k2] * LUT_WORDS);
        for (int i = (int)threadIdx.x; i < LUT_WORDS / 8; i += THREADS) {
            ((u32x4 *)lt[k2])[i] = sd[i];
            ((u32x4 *)lt[3 + k2])[i] = sa[i];
        }
    }
    if (WRITE)
        for (int i = 0; i < 4; i++) stage[i][threadIdx.x] = u32x4{0u, 0u, 0u, 0u};
    /* the MCU's block slots, packed: component (2 bits a slot), block of the component (3 bits a slot), blocks of a component (3 bits a component) */
    uint32_t cpack = 0, kbpack = 0, nbpack = 0;
    {
        uint32_t s = 0;
        for (uint32_t c = 0; c < 3; c++) {
            nbpack |= im0.nb[c] << (3 * c);
            for (uint32_t kb0 = 0; kb0 < im0.nb[c] && s < 8; kb0++, s++) {
                cpack |= c << (2 * s);
                kbpack |= kb0 << (3 * s);
            }
        }
    }
    __syncthreads();
    const uint32_t idx = i0 + threadIdx.x;
    const bool exists = idx < n_here;
    const uint32_t t = exists ? (MODE == SPAN_LIST ? a.list[idx] : idx) : t_first;
    const uint32_t p = exists ? sync_picture_of(a, t) : p0;
    const SyncImage im = a.images[p];
    bool in_lds = true;
    for (int k2 = 0; k2 < 3; k2++) in_lds = in_lds && im.tab_dc[k2] == im0.tab_dc[k2] && im.tab_ac[k2] == im0.tab_ac[k2];
    const uint32_t ti = t - a.sub_base[p];
    const uint32_t total_bits = im.clean_len * 8u;
    uint32_t limit = (ti + 1) * a.sub_bits;
    limit = limit < total_bits ? limit : total_bits;
    /* the entry: the truth for a picture's first lane, a guess in round 0, else what the lane in front recorded */
    unsigned long long entry = 0;
    if (ti > 0) {
        if (WRITE) entry = a.used[t];
        else if (a.round == 0) entry = (unsigned long long)(ti * a.sub_bits);
        else entry = __hip_atomic_load(&a.exit_[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool fresh = !WRITE && exists && (a.round == 0 || entry != a.used[t]); /* this lane's record is to be (re)written */
    bool active = WRITE ? exists : fresh;
    uint32_t pos = (uint32_t)entry, slot = (uint32_t)(entry >> 32) & 0xffu, k = (uint32_t)(entry >> 40) & 0xffu;
    if (pos >= limit) active = false; /* (an entry behind the lane's own end is handed on as it is) */
    const uint32_t nbt = im.nbt, ncomp = im.ncomp;
    u32x4 sm = {0u, 0u, 0u, 0u};
    if (WRITE && exists) sm = a.sums[t]; /* first block, DC predictors */
    slot = slot < nbt ? slot : 0;
    k = k < 64 ? k : 0;
    const uint32_t total_blocks = im.mcus * nbt;
    uint32_t bidx = 0, mcu = 0, blocks = 0;
    bool bad = false;
    if (WRITE) {
        bidx = sm[0];
        mcu = bidx / nbt;
        if (active && bidx < total_blocks && bidx - mcu * nbt != slot) { bad = true; active = false; } /* the scan and the entry disagree: not a fixed point */
        if (bidx >= total_blocks) active = false;
        mcu += im.mcu0;     /* from here on: the MCU's index in the whole plane */
    }
    /* the predictors: the current component's, the next one's in the scan's cycle, the one's after that */
    uint32_t c = (cpack >> (2 * slot)) & 3u;
    int pcur, pn1, pn2;
    {
        const int q0 = (int)sm[1], q1 = (int)sm[2], q2 = (int)sm[3];
        pcur = c == 0 ? q0 : c == 1 ? q1 : q2;
        pn1 = c == 0 ? q1 : c == 1 ? q2 : q0;
        pn2 = c == 0 ? q2 : c == 1 ? q0 : q1;
    }
    /* the bit reader: a ring of dwords per lane in LDS, (w0, w1) the two dwords in use, sh = 32 - bits of w0 taken (0..31): the next 32 bits of the
     * stream are ({w0, w1} >> sh) */
    const uint32_t start = im.scan_off + ((pos >> 5) << 2);
    const u32x4 *src = (const u32x4 *)(a.scan + (start & ~15u));
    uint32_t rd = (start & 15u) >> 2, wr = 0;
    const uint32_t dw_end = (im.scan_off + im.clean_len - (start & ~15u) + 3) >> 2;
    uint32_t *const myring = &ring[wv][0][lane];
    u32x4 pend = {0u, 0u, 0u, 0u};
    bool has_pend = false;
    auto issue = [&]() { /* (a lane with room for a chunk; behind the picture's bytes: zeros) */
        pend = u32x4{0u, 0u, 0u, 0u};
        if (wr < dw_end) pend = src[wr >> 2];
        has_pend = true;
    };
    auto commit = [&]() { /* (a lane with a chunk on its way; wr is a multiple of 4) */
        uint32_t *d = myring + (wr & (RING - 1)) * 64;
        d[0] = pend[0]; d[64] = pend[1]; d[128] = pend[2]; d[192] = pend[3];
        wr += 4;
        has_pend = false;
    };
    while (__builtin_amdgcn_ballot_w64(wr + 4 <= rd + RING)) /* to start with: the ring full */
        if (wr + 4 <= rd + RING) { issue(); commit(); }
    uint32_t w0 = 0, w1;
    int sh;
    {
        const uint32_t b = pos & 31u;
        if (b) { w0 = __builtin_bswap32(myring[(rd & (RING - 1)) * 64]); rd++; }
        w1 = __builtin_bswap32(myring[(rd & (RING - 1)) * 64]);
        rd++;
        sh = b ? 32 - (int)b : 0;
    }
    int16_t *blk = nullptr;
    auto block_of = [&]() -> int16_t * {
        const uint32_t nbc = (nbpack >> (3 * c)) & 7u, kb = (kbpack >> (3 * slot)) & 7u;
        int16_t *base = c == 0 ? a.plane[0] : c == 1 ? a.plane[1] : a.plane[2];
        return base + (size_t)(mcu * nbc + kb) * 64;
    };
    if (WRITE) blk = block_of();
    uint32_t tb_dc = c * LUT_BYTES; /* byte offset of the component's DC table in lt; its AC table 3 tables on */
    /* write pass: coefficients of the block in LDS (bit per coefficient), and whether the block is one to put together there: a block that began in the
     * lane in front, or ends in the lane behind, shares its rows with that lane and goes out coefficient by coefficient */
    uint32_t cmask = 0;
    bool staged = k == 0;
    auto flush_rows = [&]() { /* the complete block's non-zero rows, 16 bytes each */
#pragma unroll
        for (int row = 0; row < 4; row++)
            if (cmask & (0xffu << (8 * row))) {
                const u32x4 q = stage[row][threadIdx.x];
                stage[row][threadIdx.x] = u32x4{0u, 0u, 0u, 0u};
                *(u32x4 *)(blk + 8 * row) = q;
            }
        cmask = 0;
    };
    /* The loop is a plain divergent one -- a lane leaves when its subsequence is done -- and it comes in two copies: one for waves whose lanes all find
     * their tables in LDS (all but those that straddle two pictures with different tables), one with the look-ups in global memory beside it.  The only
     * global load in flight across iterations of the first is the ring's chunk: the counter waits the compiler has to place where a value MAY come from
     * memory would otherwise wait for that chunk at every symbol.  (Hence also the explicit waits at the end of the rare branches that do load.) */
    __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0): everything loaded so far is there, as far as the waits inside the loop are concerned */
    auto run = [&](auto lds_only) {
        constexpr bool LDS_ONLY = decltype(lds_only)::value;
        uint32_t iter = 1;
        while (active) {
            if ((iter & (REFILL - 1)) == 0) {
                if (has_pend) commit();
                if (wr + 4 <= rd + RING) issue();
            }
            iter++;
            const uint32_t nextdw = myring[(rd & (RING - 1)) * 64];
            const uint32_t W = __builtin_amdgcn_alignbit(w0, w1, (uint32_t)sh);
            const bool dc = k == 0;
            unsigned e;
            if (LDS_ONLY || in_lds) {
                const uint16_t *l = (const uint16_t *)((const uint8_t *)lt + tb_dc + (dc ? 0u : 3u * LUT_BYTES));
                e = l[W >> (32 - LOOK)];
                if (e & 0x8000u) e = l[512 + ((e & 0xffu) << 7) + ((W >> 16) & 127u)];
            } else {
                const uint32_t tix = dc ? (c == 0 ? im.tab_dc[0] : c == 1 ? im.tab_dc[1] : im.tab_dc[2]) : (c == 0 ? im.tab_ac[0] : c == 1 ? im.tab_ac[1] : im.tab_ac[2]);
                const uint16_t *l = a.lut + (size_t)tix * LUT_WORDS;
                e = l[W >> (32 - LOOK)];
                if (e & 0x8000u) e = l[512 + ((e & 0xffu) << 7) + ((W >> 16) & 127u)];
            }
            uint32_t sym = e & 0xffu, len = (e >> 8) & 31u;
            if (WRITE) bad |= (e & 0x4000u) != 0; /* "no such code", from a table that holds all there are */
            if (!e) { /* a code the two-level table does not hold: the canonical code, length by length */
                const uint32_t tix = dc ? (c == 0 ? im.tab_dc[0] : c == 1 ? im.tab_dc[1] : im.tab_dc[2]) : (c == 0 ? im.tab_ac[0] : c == 1 ? im.tab_ac[1] : im.tab_ac[2]);
                const struct huff *T = a.tabs + tix;
                int code = (int)(W >> (32 - LOOK)), ln = LOOK;
                while (ln < 17 && code > T->maxcode[ln]) {
                    ln++;
                    code = (int)(W >> (32 - ln));
                }
                if (ln > 16) { bad = bad || WRITE; ln = 16; code = T->mincode[16]; }
                sym = T->vals[(T->valptr[ln] + code - T->mincode[ln]) & 255];
                len = (uint32_t)ln;
                __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0) here, not behind the branch */
            }
            const uint32_t s = sym & 15u, r = dc ? 0u : sym >> 4;
            const bool skip = !dc && s == 0;
            const uint32_t used = len + s;                                        /* at most 31 */
            const uint32_t vb = __builtin_amdgcn_ubfe(W, 32u - used, s);          /* the s bits behind the code; none: 0 */
            const uint32_t m1 = (1u << s) - 1u;
            const int v = (vb << 1) > m1 ? (int)vb : (int)vb - (int)m1;          /* EXTEND, T.81 F.2.2.1 */
            pcur += dc ? v : 0; /* the predictor is a running sum: relative to the lane's start until the scan, absolute in the write pass */
            k += skip ? (r == 15 ? 16u : 64u) : r;
            if (WRITE) {
                bad |= (dc && sym > 11) || (!skip && k > 63);
                if (!skip && k <= 63) {
                    const uint32_t nat = zz[k];
                    const int16_t val = (int16_t)(dc ? pcur : v);
                    if (staged && nat < 32) { /* (a block this lane has from its first coefficient on) */
                        ((int16_t *)&stage[nat >> 3][threadIdx.x])[nat & 7u] = val;
                        cmask |= 1u << nat;
                    } else blk[nat] = val;
                }
            }
            k += skip ? 0u : 1u;
            pos += used;
            sh -= (int)used;
            if (sh < 0) { /* w0 is used up */
                w0 = w1;
                w1 = __builtin_bswap32(nextdw);
                sh += 32;
                rd++;
            }
            if (k >= 64) { /* next block: the MCU's next slot, or the next MCU */
                k = 0;
                blocks++;
                slot++;
                const bool wrap = slot == nbt;
                slot = wrap ? 0u : slot;
                const uint32_t c2 = (cpack >> (2 * slot)) & 3u;
                if (c2 != c) { const int q = pcur; pcur = pn1; pn1 = ncomp == 3 ? pn2 : q; pn2 = q; } /* (two components: they alternate) */
                c = c2;
                tb_dc = c * LUT_BYTES;
                if (WRITE) {
                    flush_rows();
                    staged = true;
                    bidx++;
                    mcu += wrap ? 1u : 0u;
                    blk = block_of();
                    if (bidx >= total_blocks) { active = false; a.end_pos[p] = pos; } /* the picture's last block */
                }
            }
            if (pos >= limit || bad) active = false;
        }
    };
    if (__builtin_amdgcn_ballot_w64(!in_lds) == 0) run(std::true_type{});
    else run(std::false_type{});
    if (WRITE) {
        while (cmask) { /* the block goes on in the lane behind: its coefficients one by one, as that lane stores them */
            const uint32_t nat = (uint32_t)__builtin_ctz(cmask);
            cmask &= cmask - 1u;
            blk[nat] = ((const int16_t *)&stage[nat >> 3][threadIdx.x])[nat & 7u];
        }
        if (bad) a.status[im.pic] = FFHIP_EINVAL;
        return;
    }
    if (fresh) {
        const int q0 = c == 0 ? pcur : c == 1 ? pn2 : pn1, q1 = c == 0 ? pn1 : c == 1 ? pcur : pn2, q2 = c == 0 ? pn2 : c == 1 ? pn1 : pcur;
        const unsigned long long ex = (unsigned long long)pos | ((unsigned long long)(slot | (k << 8)) << 32);
        __hip_atomic_store(&a.exit_[t], ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.used[t] = entry;
        u32x4 o;
        o[0] = blocks; o[1] = (uint32_t)q0; o[2] = (uint32_t)q1; o[3] = (uint32_t)q2;
        a.sums[t] = o;
    }
}

/* per picture: exclusive prefix sums over its tasks of the blocks completed and of the DC differences per component -- every task's first block and DC
 * predictors; one workgroup per picture.  A picture whose tasks hold fewer blocks than it has is truncated or damaged. */
__global__ __launch_bounds__(256) void k_huff_sync_scan(SyncArgs a)
{
    __shared__ u32x4 wsum[4];
    const uint32_t p = blockIdx.x;
    if (a.cnt[a.last]) return;
    const SyncImage im = a.images[p];
    const uint32_t t0 = a.sub_base[p], t1 = a.sub_base[p + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 carry = {0u, 0u, 0u, 0u};
    for (uint32_t base = t0; base < t1; base += 256) {
        const uint32_t t = base + threadIdx.x;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (t < t1) v = a.sums[t];
        u32x4 inc = v; /* shuffle scan inside the wave (wrapping uint32 adds: the DC sums are two's complement), the waves' totals through LDS */
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t u = (uint32_t)__shfl_up((int)inc[q], o, 64);
                if (lane >= o) inc[q] += u;
            }
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        u32x4 before = carry, total = {0u, 0u, 0u, 0u};
        for (int q = 0; q < 4; q++) {
            if (q < w) before += wsum[q];
            total += wsum[q];
        }
        if (t < t1) a.sums[t] = before + inc - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0 && carry[0] < im.mcus * im.nbt) a.status[im.pic] = FFHIP_EINVAL;
}

/* behind the write pass: a picture whose last block ends behind its data has taken bits from the padding -- truncated (the host decoder's rule,
 * ffhip_entropy.c: bytes fed behind the end are look-ahead only); one that never reached its last block left end_pos at ~0 */
__global__ __launch_bounds__(256) void k_huff_sync_verdict(SyncArgs a)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (a.cnt[a.last] || p >= a.n_images) return;
    if (a.end_pos[p] > a.images[p].data_bits) a.status[a.images[p].pic] = FFHIP_EINVAL;
}

namespace {
struct SyncLayout { size_t o_img, o_sub, o_cnt, o_end, o_list, o_seg, o_sums, o_exit, o_used, words; };
SyncLayout sync_layout(size_t n, size_t tasks)
{
    SyncLayout L;
    size_t w = 0;
    auto take = [&](size_t words) { const size_t at = w; w += (words + 3) & ~(size_t)3; return at; };
    L.o_img = take(n * sizeof(SyncImage) / 4);
    L.o_sub = take(n + 1);
    L.o_cnt = take(SYNC_ROUNDS_MAX + 3);
    L.o_end = take(n);
    L.o_list = take(tasks);
    L.o_seg = take(tasks);
    L.o_sums = take(4 * tasks);
    L.o_exit = take(2 * tasks);
    L.o_used = take(2 * tasks);
    L.words = w;
    return L;
}
void sync_args(SyncArgs &a, const SyncJob &job, uint32_t *d, const SyncLayout &L)
{
    a.scan = job.dev;
    a.tabs = (const struct huff *)(job.dev + job.o_tabs);
    a.lut = (const uint16_t *)(job.dev + job.o_l12);
    a.images = (const SyncImage *)(d + L.o_img);
    a.sub_base = d + L.o_sub;
    a.exit_ = (unsigned long long *)(d + L.o_exit);
    a.used = (unsigned long long *)(d + L.o_used);
    a.sums = (u32x4 *)(d + L.o_sums);
    a.list = d + L.o_list;
    a.seg_of = d + L.o_seg;
    a.cnt = d + L.o_cnt;
    a.end_pos = d + L.o_end;
    for (int c = 0; c < 3; c++) a.plane[c] = job.plane[c];
    a.status = (int *)(job.dev + job.o_status);
    a.n_tasks = job.n_tasks;
    a.n_images = (uint32_t)job.n_segs;
    a.round = 0;
    a.last = job.rounds + 2;
    a.sub_bits = job.sub_bits;
}
uint32_t sync_tasks_of(uint32_t raw_len, uint32_t sub_bits)
{
    const uint64_t bits = (uint64_t)raw_len * 8;
    return bits ? (uint32_t)((bits + sub_bits - 1) / sub_bits) : 1u;
}
/* list rounds 2 .. rounds + 1, the list of round rounds + 2 (empty: the fixed point), then the passes that need it */
int sync_launch_rounds_and_passes(SyncArgs &a, const SyncJob &job, hipStream_t st)
{
    const unsigned wgs = (job.n_tasks + SPAN_THREADS - 1) / SPAN_THREADS, wgl = (job.n_tasks + LIST_THREADS - 1) / LIST_THREADS;
    for (uint32_t r = 2; r <= job.rounds + 2; r++) {
        a.round = r;
        hipLaunchKernelGGL(k_huff_sync_list, dim3(wgl), dim3(LIST_THREADS), 0, st, a);
        if (r < job.rounds + 2) hipLaunchKernelGGL((k_huff_span<SPAN_LIST, SPAN_THREADS>), dim3(wgs), dim3(SPAN_THREADS), 0, st, a);
    }
    hipLaunchKernelGGL(k_huff_sync_scan, dim3((unsigned)job.n_segs), dim3(256), 0, st, a);
    hipLaunchKernelGGL((k_huff_span<SPAN_WRITE, WRITE_THREADS>), dim3((job.n_tasks + WRITE_THREADS - 1) / WRITE_THREADS), dim3(WRITE_THREADS), 0, st, a);
    hipLaunchKernelGGL(k_huff_sync_verdict, dim3(((unsigned)job.n_segs + 255) / 256), dim3(256), 0, st, a);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
} // namespace

/* everything of the subsequence decoder on the call's stream: picture records up, rounds, scan, write pass, the rounds' verdict back */
static int huff_sync_enqueue(SyncJob &job, void *stream, uint32_t **h_cnt)
{
    static_assert(sizeof(SyncImage) % 4 == 0, "SyncImage is copied by words");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = job.n_segs; /* "pictures" of the kernels: the intervals */
    {
        const char *e = FFHIP_ENV("FFHIP_JPEG_SYNC_ROUNDS");
        const int r = e ? atoi(e) : 10;
        job.rounds = (uint32_t)(r < 1 ? 1 : r > SYNC_ROUNDS_MAX ? SYNC_ROUNDS_MAX : r);
        /* bits of a subsequence (job.sub_bits, from sync_sub_bits): by the bits an MCU takes in the batch, unless FFHIP_JPEG_SYNC_BITS sets them.  A wrong block state is put right over a few MCUs, and every subsequence it survives is
         * a round -- of a sparse kernel that takes as long as one lane takes for its subsequence: 128 4K files of quality 95 (1 500 bits an MCU) 11 rounds
         * and 29 ms at 2 048 bits, 6 rounds and 24 ms at 4 096; quality 100 on noise (2 800 bits an MCU, hardly an end-of-block anywhere) 59 rounds and
         * 91 ms at 2 048, 16 and 71 at 8 192.  Longer subsequences than that lose more in the sparse rounds than they save in their number. */
        if (job.sub_bits < 128 || job.sub_bits > 65536) return FFHIP_EINVAL;
    }
    uint64_t tasks = 0;
    for (size_t i = 0; i < n; i++) {
        if (job.segs[i].clean_len >= (1u << 28)) return FFHIP_EINVAL; /* (bit positions are 32-bit words: an interval of a quarter of a gigabyte goes to the host decoder) */
        tasks += sync_tasks_of(job.segs[i].raw_len, job.sub_bits);
    }
    if (tasks > 0x7fffff00u || n > 0x7fffff00u) return FFHIP_EINVAL;
    const SyncLayout L = sync_layout(n, (size_t)tasks);
    const size_t head = L.o_list; /* words the host fills: interval records, first tasks, list counts, end positions */
    uint8_t *pin = ffhip_pinned_scratch(SCRATCH_HUFF_SYNC + job.part, stream, (head + SYNC_ROUNDS_MAX + 3) * 4);
    uint32_t *d = ffhip_scratch(SCRATCH_HUFF_SYNC + job.part, stream, L.words);
    if (!pin || !d) return FFHIP_ENOMEM;
    uint32_t *h = (uint32_t *)pin;
    SyncImage *si = (SyncImage *)(h + L.o_img);
    uint32_t t = 0;
    for (size_t i = 0; i < n; i++) {
        const SyncSeg &sg = job.segs[i];
        const HuffImage &im = job.images[sg.pic];
        SyncImage &s = si[i];
        s.scan_off = sg.scan_off;
        s.clean_len = sg.clean_len;
        s.data_bits = sg.raw_len * 8u;
        s.mcus = sg.mcus;
        s.pic = sg.pic;
        s.mcu0 = sg.mcu0;
        s.ncomp = im.ncomp;
        s.nbt = 0;
        for (int c = 0; c < 3; c++) { s.nb[c] = im.nb[c]; s.tab_dc[c] = im.tab_dc[c]; s.tab_ac[c] = im.tab_ac[c]; s.nbt += im.nb[c]; }
        if (s.nb[0] != si[0].nb[0] || s.nb[1] != si[0].nb[1] || s.nb[2] != si[0].nb[2]) return FFHIP_EINVAL; /* (one geometry per call: the kernels keep one record of the MCU's blocks) */
        h[L.o_sub + i] = t;
        t += sync_tasks_of(sg.raw_len, job.sub_bits);
    }
    h[L.o_sub + n] = t;
    job.n_tasks = t;
    memset(h + L.o_cnt, 0, (L.o_end - L.o_cnt) * 4);
    memset(h + L.o_end, 0xff, (L.o_list - L.o_end) * 4);
    FFHIP_CHECK(hipMemcpyAsync(d, h, head * 4, hipMemcpyHostToDevice, st), FFHIP_EIO);
    SyncArgs a;
    sync_args(a, job, d, L);
    const unsigned wgs = (t + SPAN_THREADS - 1) / SPAN_THREADS;
    hipLaunchKernelGGL(k_huff_sync_segs, dim3((unsigned)n), dim3(256), 0, st, a);
    for (uint32_t r = 0; r < 2; r++) {
        a.round = r;
        hipLaunchKernelGGL((k_huff_span<SPAN_ALL, SPAN_THREADS>), dim3(wgs), dim3(SPAN_THREADS), 0, st, a);
    }
    const int rc = sync_launch_rounds_and_passes(a, job, st);
    if (rc) return rc;
    *h_cnt = h + head;
    FFHIP_CHECK(hipMemcpyAsync(*h_cnt, d + L.o_cnt, (SYNC_ROUNDS_MAX + 3) * 4, hipMemcpyDeviceToHost, st), FFHIP_EIO);
    return FFHIP_OK;
}

/* behind the stream's sync.  The rounds reached their fixed point: nothing to do.  They did not (a scan built so that a wrong start stays wrong over many
 * subsequences -- no encoder's output): further rounds until they do, at most as many as the longest picture has subsequences, then the passes again. */
static int huff_sync_finish(SyncJob &job, void *stream, uint32_t *h_cnt, int *status)
{
    hipStream_t st = (hipStream_t)stream;
    const uint32_t last = job.rounds + 2;
    job.rounds_used = 2;
    job.reran = 0;
    for (uint32_t r = 2; r < last; r++) job.rounds_used += h_cnt[r] ? 1u : 0u;
    if (!h_cnt[last]) return FFHIP_OK;
    const size_t n = job.n_segs;
    const SyncLayout L = sync_layout(n, job.n_tasks);
    uint32_t *d = ffhip_scratch(SCRATCH_HUFF_SYNC + job.part, stream, L.words);
    if (!d) return FFHIP_ENOMEM;
    SyncArgs a;
    sync_args(a, job, d, L);
    uint32_t longest = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t ti = sync_tasks_of(job.segs[i].raw_len, job.sub_bits);
        longest = ti > longest ? ti : longest;
    }
    for (uint32_t done = last;; done += job.rounds) {
        if (done > longest + last) return FFHIP_EIO; /* (cannot happen: lane t is final after round t) */
        memset(h_cnt, 0, (SYNC_ROUNDS_MAX + 3) * 4);
        FFHIP_CHECK(hipMemcpyAsync(d + L.o_cnt, h_cnt, (SYNC_ROUNDS_MAX + 3) * 4, hipMemcpyHostToDevice, st), FFHIP_EIO);
        const int rc = sync_launch_rounds_and_passes(a, job, st); /* (the passes look at the last list themselves) */
        if (rc) return rc;
        FFHIP_CHECK(hipMemcpyAsync(h_cnt, d + L.o_cnt, (SYNC_ROUNDS_MAX + 3) * 4, hipMemcpyDeviceToHost, st), FFHIP_EIO);
        FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
        for (uint32_t r = 2; r < last; r++) job.rounds_used += h_cnt[r] ? 1u : 0u;
        if (!h_cnt[last]) break;
    }
    FFHIP_CHECK(hipMemcpy(status, job.dev + job.o_status, (size_t)job.n * 4, hipMemcpyDeviceToHost), FFHIP_EIO);
    job.reran = 1;
    return FFHIP_OK;
}
