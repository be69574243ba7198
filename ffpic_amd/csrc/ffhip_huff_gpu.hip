/*
 * ffhip_huff_gpu.hip -- the JPEG entropy front end ON the GPU for files that carry restart markers
 * ("Huffman is serial per restart interval" -- so the intervals are the parallelism).
 *
 * What the host still does per file: the marker loop and table building of ffhip_entropy.c
 * (format/jpg.c:78-105, 640-655, 771-855) and a scan for the RSTn markers.  What moves to the device:
 * read_compressed_scan / decode_data_unit (format/jpg.c:255-415, 562-573, 588-637) -- ONE LANE per
 * restart interval walks its bytes (FF00 unstuffing, 64-bit bit buffer), decodes symbol after symbol with
 * the same 9-bit look-up tables the host decoder uses (struct huff, uploaded as is) and scatters the
 * coefficients, de-zigzagged, into the MCU-order planes ffhip_jpeg_recon_batch reads.  The decode loop is
 * a flat one-symbol-per-iteration state machine so that lanes in different blocks, components or MCUs
 * still execute the same instructions.  Files without a DRI segment are refused (FFHIP_EINVAL): they
 * have one interval, i.e. no parallelism to offer, and stay on the host threads.
 */
#include "ffhip_internal.h"
#include "ffhip_entropy_internal.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>

struct HuffImage {
    uint32_t scan_off, scan_len; /* this picture's entropy-coded bytes inside `scan` */
    uint32_t seg_base, n_seg;    /* its restart intervals inside `seg` (offsets relative to scan_off) */
    uint32_t restart, mcus;
    uint32_t ncomp, nb[3];       /* blocks per MCU and component */
    uint32_t tab_dc[3], tab_ac[3]; /* indices into `tabs` */
};

struct HuffArgs {
    const uint8_t *scan;
    const struct huff *tabs;
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

__global__ __launch_bounds__(64) void k_jpeg_huff(HuffArgs a)
{
    __shared__ uint8_t zz[64];
    /* the 9-bit tables of the wave's FIRST picture in LDS (a wave's 64 intervals belong to one picture, rarely two):
     * [0..2] DC look-ups, [3..5] AC look-ups, per component.  (The host decoder's one-look-up run/value path for
     * small AC coefficients was tried here and lost: lanes that take it and lanes that do not serialise.) */
    __shared__ uint16_t lt[6][1 << LOOK];
    zz[threadIdx.x] = kZigzag[threadIdx.x];
    const uint32_t gid0 = blockIdx.x * 64;
    const uint32_t img0 = a.work[gid0].x; /* gid0 < n_work: the grid is not larger than the work list */
    {
        const HuffImage im0 = a.images[img0];
        for (int t = 0; t < 3; t++)
            for (int i = threadIdx.x; i < (1 << LOOK); i += 64) {
                lt[t][i] = a.tabs[im0.tab_dc[t]].look[i];
                lt[3 + t][i] = a.tabs[im0.tab_ac[t]].look[i];
            }
    }
    __syncthreads();
    const uint32_t gid = gid0 + threadIdx.x;
    if (gid >= a.n_work) return;
    const u32x2 w = a.work[gid];
    const HuffImage im = a.images[w.x];
    const bool in_lds = w.x == img0;
    const uint8_t *base = a.scan + im.scan_off;
    uint32_t p = a.seg[im.seg_base + w.y];
    const uint32_t end = w.y + 1 < im.n_seg ? a.seg[im.seg_base + w.y + 1] - 2 : im.scan_len; /* stop in front of the RSTn */
    uint32_t mcu = w.y * im.restart;
    const uint32_t mcu_end = mcu + im.restart < im.mcus ? mcu + im.restart : im.mcus;
    /* per component: plane pointer of this picture, tables */
    int16_t *pl[3];
    const struct huff *tdc[3], *tac[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        pl[c] = a.plane[c] ? a.plane[c] + (size_t)w.x * im.mcus * im.nb[c] * 64 : nullptr;
        tdc[c] = a.tabs + im.tab_dc[c];
        tac[c] = a.tabs + im.tab_ac[c];
    }
    unsigned long long acc = 0;
    int n = 0;
    bool marker = false, bad = false;
    int pred[3] = {0, 0, 0};
    uint32_t c = 0, kb = 0, k = 0;
    while (mcu < mcu_end) {
        if (n < 32) { /* refill to more than 56 bits: 8 plain bytes at once, or byte by byte around 0xFF */
            if (!marker && p + 8 <= end) {
                const unsigned long long lo = *(const unsigned int *)(base + p), hi = *(const unsigned int *)(base + p + 4); /* unaligned global loads */
                const unsigned long long le = lo | (hi << 32);
                const unsigned long long be = ((unsigned long long)__builtin_bswap32((unsigned)le) << 32) | __builtin_bswap32((unsigned)(le >> 32));
                const unsigned long long x = ~be;
                if (!((x - 0x0101010101010101ULL) & ~x & 0x8080808080808080ULL)) {
                    const int nb = (64 - n) >> 3;
                    acc = nb == 8 ? be : (acc << (8 * nb)) | (be >> (64 - 8 * nb));
                    p += nb;
                    n += 8 * nb;
                }
            }
            while (n <= 56) {
                unsigned cbyte = 0;
                if (!marker && p < end) {
                    cbyte = base[p];
                    if (cbyte == 0xFF) {
                        if (p + 1 < end && base[p + 1] == 0) p += 2; /* stuffed zero (jpg.c:588-637) */
                        else { marker = true; cbyte = 0; }
                    } else p++;
                }
                acc = (acc << 8) | cbyte;
                n += 8;
            }
        }
        const uint32_t cc = c;
        const struct huff *T = k == 0 ? (cc == 0 ? tdc[0] : (cc == 1 ? tdc[1] : tdc[2])) : (cc == 0 ? tac[0] : (cc == 1 ? tac[1] : tac[2]));
        const unsigned peek = (unsigned)(acc >> (n - LOOK)) & ((1u << LOOK) - 1);
        /* one Huffman symbol (coding/huffman.c:92-222): 9-bit look-up, then the canonical-code walk */
        int sym;
        {
            const unsigned e = in_lds ? lt[(k == 0 ? 0 : 3) + cc][peek] : T->look[peek];
            if (e) { n -= (int)(e >> 8); sym = (int)(e & 0xff); }
            else {
                int code = (int)peek, len = LOOK;
                while (len < 17 && code > T->maxcode[len]) {
                    len++;
                    code = (int)((acc >> (n - len)) & ((1u << len) - 1));
                }
                if (len > 16) { bad = true; break; }
                n -= len;
                sym = T->vals[(T->valptr[len] + code - T->mincode[len]) & 255]; /* & 255: a malformed DHT must not index outside the table */
            }
        }
        const bool dc = k == 0;
        const int s = dc ? sym : (sym & 15), r = dc ? 0 : (sym >> 4);
        if (dc && s > 11) { bad = true; break; }
        if (!dc && s == 0) {
            k = r == 15 ? k + 16 : 64; /* ZRL / EOB */
        } else {
            k += (uint32_t)r;
            if (k > 63) { bad = true; break; }
            int v = 0;
            if (s) {
                v = (int)((acc >> (n - s)) & ((1u << s) - 1));
                n -= s;
                if (v < (1 << (s - 1))) v -= (1 << s) - 1; /* EXTEND, T.81 F.2.2.1 */
            }
            if (dc) {
                pred[0] = cc == 0 ? pred[0] + v : pred[0];
                pred[1] = cc == 1 ? pred[1] + v : pred[1];
                pred[2] = cc == 2 ? pred[2] + v : pred[2];
                v = cc == 0 ? pred[0] : (cc == 1 ? pred[1] : pred[2]);
            }
            int16_t *P = cc == 0 ? pl[0] : (cc == 1 ? pl[1] : pl[2]);
            const uint32_t nbc = cc == 0 ? im.nb[0] : (cc == 1 ? im.nb[1] : im.nb[2]);
            P[((size_t)mcu * nbc + kb) * 64 + zz[k]] = (int16_t)v;
            k++;
        }
        if (k >= 64) { /* next block of the MCU, next component, next MCU */
            k = 0;
            const uint32_t nbc = cc == 0 ? im.nb[0] : (cc == 1 ? im.nb[1] : im.nb[2]);
            if (++kb == nbc) {
                kb = 0;
                if (++c == im.ncomp) { c = 0; mcu++; }
            }
        }
    }
    if (bad) a.status[w.x] = FFHIP_EINVAL;
}

#define SCRATCH_HUFF 4

namespace {
/* pinned staging for the one upload per call, kept and grown; calls are serialised (they end in a stream sync) */
uint8_t *g_stage = nullptr;
size_t g_stage_cap = 0;
std::mutex g_huff_mu;

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

extern "C" int ffhip_jpeg_entropy_batch_gpu(const uint8_t *const *files, const size_t *lens, int n, int n_threads,
                                            const ffhip_jpeg_geom *geom, int16_t *d_coef_y, int16_t *d_coef_u, int16_t *d_coef_v,
                                            uint16_t *d_quant, int *status, void *stream)
{
    if (n < 0 || !geom || (n > 0 && (!files || !lens || !d_coef_y || !d_quant || !status))) return FFHIP_EINVAL;
    if (n == 0) return FFHIP_OK;
    if (geom->ncomp == 3 && (!d_coef_u || !d_coef_v)) return FFHIP_EINVAL;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    const size_t mcus = (size_t)geom->mcu_cols * geom->mcu_rows;
    /* ---- host, pictures over threads: headers, tables, restart-interval starts ---- */
    std::vector<struct jpeg_hdr> hdr((size_t)n);
    std::vector<std::vector<uint32_t>> segs((size_t)n);
    parallel_for(n, n_threads, [&](int i) {
        struct jpeg_hdr &j = hdr[(size_t)i];
        status[i] = ffhip_jpeg_parse(files[i], lens[i], &j);
        if (status[i]) return;
        const int mc = (j.width + 8 * j.h[0] - 1) / (8 * j.h[0]), mr = (j.height + 8 * j.v[0] - 1) / (8 * j.v[0]);
        if (mc != geom->mcu_cols || mr != geom->mcu_rows || j.ncomp != geom->ncomp || j.h[0] != geom->h || j.v[0] != geom->v || !j.restart ||
            j.scan_len > 0x7fffffffu) {
            status[i] = FFHIP_EINVAL; /* another geometry, or no restart intervals to spread over lanes */
            return;
        }
        /* interval starts: behind the RSTn markers (0xFF is stuffed inside entropy data, so FF D0..D7 is a marker) */
        const uint32_t n_seg = (uint32_t)((mcus + j.restart - 1) / j.restart);
        std::vector<uint32_t> &sg = segs[(size_t)i];
        sg.reserve(n_seg);
        sg.push_back(0);
        const uint8_t *s = j.scan, *e = j.scan + j.scan_len;
        for (const uint8_t *q = s; sg.size() < n_seg && (q = (const uint8_t *)memchr(q, 0xFF, (size_t)(e - q))) != nullptr && q + 1 < e; q++)
            if (q[1] >= 0xD0 && q[1] <= 0xD7) { sg.push_back((uint32_t)(q + 2 - s)); q++; }
        if (sg.size() != n_seg) status[i] = FFHIP_EINVAL;
    });
    for (int i = 0; i < n; i++)
        if (status[i]) return status[i];
    /* ---- layout of the one upload: scan bytes | tables | picture records | interval starts | work list | status ---- */
    std::vector<HuffImage> images((size_t)n);
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
            im.tab_dc[c] = (uint32_t)(i * 6 + 2 * c);
            im.tab_ac[c] = (uint32_t)(i * 6 + 2 * c + 1);
        }
        scan_total += (j.scan_len + 15) & ~(size_t)15;
        seg_total += im.n_seg;
        if (scan_total > 0x7fffffffu) return FFHIP_EINVAL;
    }
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    const size_t o_tabs = scan_total + 16, o_img = (o_tabs + (size_t)n * 6 * sizeof(struct huff) + 15) & ~(size_t)15;
    const size_t o_seg = (o_img + images.size() * sizeof(HuffImage) + 15) & ~(size_t)15, o_work = (o_seg + seg_total * 4 + 15) & ~(size_t)15;
    const size_t o_status = (o_work + seg_total * 8 + 15) & ~(size_t)15, o_quant = (o_status + (size_t)n * 4 + 15) & ~(size_t)15;
    const size_t total = o_quant + (size_t)n * 512;
    std::lock_guard<std::mutex> lock(g_huff_mu);
    if (total > g_stage_cap) {
        if (g_stage) (void)hipHostFree(g_stage);
        g_stage = nullptr;
        g_stage_cap = 0;
        const size_t want = total + total / 4;
        if (hipHostMalloc((void **)&g_stage, want, hipHostMallocDefault) != hipSuccess) { g_stage = nullptr; return FFHIP_ENOMEM; }
        g_stage_cap = want;
    }
    uint8_t *stage = g_stage;
    parallel_for(n, n_threads, [&](int i) {
        const struct jpeg_hdr &j = hdr[(size_t)i];
        const HuffImage &im = images[(size_t)i];
        memcpy(stage + im.scan_off, j.scan, j.scan_len);
        memset(stage + im.scan_off + j.scan_len, 0, (((j.scan_len + 15) & ~(size_t)15) - j.scan_len));
        struct huff *tb = (struct huff *)(stage + o_tabs) + (size_t)i * 6;
        for (int c = 0; c < 3; c++) {
            tb[2 * c] = j.dc[c < j.ncomp ? j.td[c] : j.td[0]];
            tb[2 * c + 1] = j.ac[c < j.ncomp ? j.ta[c] : j.ta[0]];
        }
        uint32_t *sg = (uint32_t *)(stage + o_seg) + im.seg_base;
        u32x2 *wk = (u32x2 *)(stage + o_work) + im.seg_base;
        for (uint32_t k = 0; k < im.n_seg; k++) {
            sg[k] = segs[(size_t)i][k];
            wk[k].x = (uint32_t)i;
            wk[k].y = k;
        }
        memcpy(stage + o_quant + (size_t)i * 512, j.quant, 512);
    });
    memset(stage + scan_total, 0, 16);
    memcpy(stage + o_img, images.data(), images.size() * sizeof(HuffImage));
    memset(stage + o_status, 0, (size_t)n * 4);
    hipStream_t st = (hipStream_t)stream;
    uint8_t *dev = (uint8_t *)ffhip_scratch(SCRATCH_HUFF, stream, (total + 3) / 4);
    if (!dev) return FFHIP_ENOMEM;
    FFHIP_CHECK(hipMemcpyAsync(dev, stage, total, hipMemcpyHostToDevice, st), FFHIP_EIO); /* ordered behind this stream's earlier batch */
    FFHIP_CHECK(hipMemcpyAsync(d_quant, dev + o_quant, (size_t)n * 512, hipMemcpyDeviceToDevice, st), FFHIP_EIO);
    const size_t yb = mcus * geom->h * geom->v * 64;
    FFHIP_CHECK(hipMemsetAsync(d_coef_y, 0, (size_t)n * yb * 2, st), FFHIP_EIO);
    if (geom->ncomp == 3) {
        FFHIP_CHECK(hipMemsetAsync(d_coef_u, 0, (size_t)n * mcus * 128, st), FFHIP_EIO);
        FFHIP_CHECK(hipMemsetAsync(d_coef_v, 0, (size_t)n * mcus * 128, st), FFHIP_EIO);
    }
    HuffArgs a;
    a.scan = dev;
    a.tabs = (const struct huff *)(dev + o_tabs);
    a.images = (const HuffImage *)(dev + o_img);
    a.seg = (const uint32_t *)(dev + o_seg);
    a.work = (const u32x2 *)(dev + o_work);
    a.plane[0] = d_coef_y; a.plane[1] = d_coef_u; a.plane[2] = d_coef_v;
    a.status = (int *)(dev + o_status);
    a.n_work = (uint32_t)seg_total;
    hipLaunchKernelGGL(k_jpeg_huff, dim3((a.n_work + 63) / 64), dim3(64), 0, st, a);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    /* per-picture verdicts come back with the stream (tiny); the staging buffer is free again after this sync */
    FFHIP_CHECK(hipMemcpyAsync(stage + o_status, dev + o_status, (size_t)n * 4, hipMemcpyDeviceToHost, st), FFHIP_EIO);
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO);
    memcpy(status, stage + o_status, (size_t)n * 4);
    for (int i = 0; i < n; i++)
        if (status[i]) return status[i];
    return FFHIP_OK;
}
