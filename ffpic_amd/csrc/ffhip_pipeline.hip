/*
 * ffhip_pipeline.hip -- files in, pixels out: the JPEG front end (ffhip_entropy.c, host threads) and the
 * fused reconstruction (ffhip_jpeg.hip) as one double-buffered pipeline, the transbmp-shaped caller of
 * SURVEY 8 rows f1 + f2 (format/jpg.c:588-655 -> :540-560 -> struct pic, format/file.h:29-40).
 *
 * Pictures of one geometry are processed in chunks.  While the host threads Huffman-decode chunk k + 1
 * straight into pinned memory, chunk k is on its stream: H2D copy, one reconstruction launch, D2H copy into
 * pinned memory; finished chunks are copied out to the caller's (pageable) buffer.  Steady state is the
 * slower of "entropy decode on the host" and "PCIe", not their sum.  PCIe-inclusive by construction: this
 * is never the number bench.py reports.
 */
#include "ffhip_internal.h"
#include "ffhip_entropy_internal.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>

namespace {
/* two slots of pinned + device buffers and a stream each, kept between calls (pinning hundreds of MB costs
 * more than decoding them) and grown on demand; one pipeline call at a time (mutex) */
struct Slot {
    int16_t *h_y = nullptr, *h_u = nullptr, *h_v = nullptr; /* pinned */
    uint16_t *h_q = nullptr;
    uint8_t *h_out = nullptr;                               /* pinned staging, unused when the caller's buffer is pinned */
    int16_t *d_y = nullptr, *d_u = nullptr, *d_v = nullptr;
    uint16_t *d_q = nullptr;
    uint8_t *d_out = nullptr;
    hipStream_t st = nullptr;
    size_t cap_y = 0, cap_c = 0, cap_q = 0, cap_out = 0, cap_hout = 0; /* bytes */
    int first = -1, count = 0; /* pictures in flight in this slot */
};
Slot g_slot[2];
std::mutex g_pipe_mu;

bool grow_pair(void **h, void **d, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return true;
    if (*h) (void)hipHostFree(*h);
    if (*d) (void)hipFree(*d);
    *h = *d = nullptr;
    *cap = 0;
    if (hipHostMalloc(h, bytes, hipHostMallocDefault) != hipSuccess) { *h = nullptr; return false; }
    if (hipMalloc(d, bytes) != hipSuccess) { *d = nullptr; return false; }
    *cap = bytes;
    return true;
}
bool prepare(Slot &s, size_t by, size_t bc, size_t bq, size_t bout, bool need_hout)
{
    if (!s.st && hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) return false;
    if (!grow_pair((void **)&s.h_y, (void **)&s.d_y, &s.cap_y, by)) return false;
    if (bc > s.cap_c) {
        size_t c1 = s.cap_c, c2 = s.cap_c;
        if (!grow_pair((void **)&s.h_u, (void **)&s.d_u, &c1, bc) || !grow_pair((void **)&s.h_v, (void **)&s.d_v, &c2, bc)) { s.cap_c = 0; return false; }
        s.cap_c = bc;
    }
    if (!grow_pair((void **)&s.h_q, (void **)&s.d_q, &s.cap_q, bq)) return false;
    if (bout > s.cap_out) {
        if (s.d_out) (void)hipFree(s.d_out);
        s.d_out = nullptr; s.cap_out = 0;
        if (hipMalloc((void **)&s.d_out, bout) != hipSuccess) { s.d_out = nullptr; return false; }
        s.cap_out = bout;
    }
    if (need_hout && bout > s.cap_hout) {
        if (s.h_out) (void)hipHostFree(s.h_out);
        s.h_out = nullptr; s.cap_hout = 0;
        if (hipHostMalloc((void **)&s.h_out, bout, hipHostMallocDefault) != hipSuccess) { s.h_out = nullptr; return false; }
        s.cap_hout = bout;
    }
    s.first = -1;
    s.count = 0;
    return true;
}
/* rows of `count` pictures from tight pinned staging to the caller's pageable buffer, over host threads */
void copy_out(uint8_t *bgra, int64_t pitch, int64_t image_stride, const uint8_t *src, size_t dev_pitch, int64_t height, int first,
              int count, int n_threads)
{
    const long long rows = (long long)count * height;
    auto part = [&](int t, int nt) {
        for (long long r = rows * t / nt, e = rows * (t + 1) / nt; r < e; r++) {
            const long long i = r / height, y = r % height;
            memcpy(bgra + (first + i) * image_stride + y * pitch, src + (size_t)r * dev_pitch, dev_pitch);
        }
    };
    const int nt = n_threads < 1 ? 1 : (n_threads > 16 ? 16 : n_threads);
    if (nt == 1 || rows < 64) { part(0, 1); return; }
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(part, t, nt);
    part(0, nt);
    for (auto &th : pool) th.join();
}
} // namespace

extern "C" void ffhip_pipeline_release(void)
{
    std::lock_guard<std::mutex> lock(g_pipe_mu);
    for (int s = 0; s < 2; s++) {
        Slot &sl = g_slot[s];
        if (sl.st) (void)hipStreamSynchronize(sl.st);
        (void)hipHostFree(sl.h_y); (void)hipHostFree(sl.h_u); (void)hipHostFree(sl.h_v); (void)hipHostFree(sl.h_q); (void)hipHostFree(sl.h_out);
        (void)hipFree(sl.d_y); (void)hipFree(sl.d_u); (void)hipFree(sl.d_v); (void)hipFree(sl.d_q); (void)hipFree(sl.d_out);
        if (sl.st) (void)hipStreamDestroy(sl.st);
        sl = Slot();
    }
}

extern "C" void *ffhip_host_malloc(size_t bytes)
{
    void *p = nullptr;
    if (!ffhip_have_device()) return nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void ffhip_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int ffhip_jpeg_decode_files(const uint8_t *const *files, const size_t *lens, int n, int n_threads, int chunk,
                                       ffhip_jpeg_geom *geom_out, uint8_t *bgra, int64_t pitch, int64_t image_stride,
                                       int *status)
{
    if (n < 0 || (n > 0 && (!files || !lens || !bgra || !status))) return FFHIP_EINVAL;
    if (n == 0) return FFHIP_OK;
    ffhip_jpeg_geom g;
    int w = 0, h = 0;
    int rc = ffhip_jpeg_probe(files[0], lens[0], &g, &w, &h);
    if (rc) return rc;
    if (geom_out) *geom_out = g;
    const int64_t width = (int64_t)g.mcu_cols * 8 * g.h, height = (int64_t)g.mcu_rows * 8 * g.v;
    if (pitch < width * 4 || (pitch & 15) || (n > 1 && image_stride < pitch * height)) return FFHIP_EINVAL;
    if (g.mcu_cols <= 0 || g.mcu_rows <= 0) return FFHIP_EINVAL; /* workspace_bytes is 0 for a geometry it rejects, too */
    if (ffhip_jpeg_workspace_bytes(&g, 1) != 0) return FFHIP_EINVAL; /* one component with several blocks per MCU: not here */
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    /* the entropy decode runs on the device (FFHIP_JPEG_GPU_ENTROPY=0 keeps it off): the subsequence decoder, whatever the files' restart markers; with
     * FFHIP_JPEG_SYNC=0, round 4's kernel -- a lane per restart interval, its latency per batch that of ONE interval --, for files that have markers */
    const char *ge = FFHIP_ENV("FFHIP_JPEG_GPU_ENTROPY");
    bool gpu_entropy = !(ge && ge[0] == '0');
    if (gpu_entropy && !(ge && ge[0] == '1')) { /* "1" forces it; default: always, unless files without restart markers are to be one lane each (FFHIP_JPEG_SYNC=0) */
        const char *sy = FFHIP_ENV("FFHIP_JPEG_SYNC");
        gpu_entropy = !(sy && sy[0] == '0') || ffhip_jpeg_probe_restart(files[0], lens[0]) > 0 || n >= 1024;
    }
    if (chunk <= 0) {
        /* a chunk is a device call and a stream sync: 32 pictures of 4K (a gigabyte of BGRA per slot), and as many small pictures as make 256 MB of BGRA
         * -- 1 024 thumbnails of 256x256, not 32 */
        chunk = gpu_entropy ? 32 : 8;
        const int64_t px = width * height * 4;
        if (gpu_entropy && px > 0 && (256ll << 20) / px > chunk) chunk = (int)((256ll << 20) / px > 4096 ? 4096 : (256ll << 20) / px);
    }
    if (chunk > n) chunk = n;
    const size_t mcus = (size_t)g.mcu_cols * g.mcu_rows;
    const size_t yb = mcus * g.h * g.v * 64, cb = g.ncomp == 3 ? mcus * 64 : 0; /* int16 elements per picture */
    /* on the device the pictures have the pitch ffhip_bgra_layout recommends (the buffer is the library's); the pinned staging
     * for a pageable destination is tight, and every copy off the device is a 2-D copy */
    int64_t lp = 0, ls = 0;
    if (ffhip_bgra_layout(&g, &lp, &ls) != FFHIP_OK) return FFHIP_EINVAL;
    const size_t dev_pitch = (size_t)lp, out_b = (size_t)ls, row_b = (size_t)width * 4;
    /* a pinned (hipHostMalloc'ed / registered) destination takes the D2H copy directly */
    hipPointerAttribute_t attr;
    const bool pinned_dst = hipPointerGetAttributes(&attr, bgra) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned_dst) (void)hipGetLastError(); /* an unknown pointer leaves an error behind: not ours */

    std::lock_guard<std::mutex> lock(g_pipe_mu);
    Slot *slot = g_slot;
    for (int s = 0; s < 2; s++)
        if (!prepare(slot[s], chunk * yb * 2, chunk * cb * 2, (size_t)chunk * 512, chunk * out_b, !pinned_dst)) return FFHIP_ENOMEM;
    int result = FFHIP_OK;
    /* wait for a slot's chunk and hand its pixels to the caller */
    auto drain = [&](Slot &sl) -> int {
        if (sl.count == 0) return FFHIP_OK;
        if (hipStreamSynchronize(sl.st) != hipSuccess) return FFHIP_EIO;
        if (!pinned_dst) copy_out(bgra, pitch, image_stride, sl.h_out, row_b, height, sl.first, sl.count, n_threads);
        sl.count = 0;
        return FFHIP_OK;
    };
    rc = FFHIP_OK;
    int k = 0;
    for (int first = 0; first < n && rc == FFHIP_OK; first += chunk, k++) {
        Slot &sl = slot[k & 1];
        const int cnt = n - first < chunk ? n - first : chunk;
        rc = drain(sl); /* the slot's previous chunk (k - 2) */
        if (rc) break;
        hipError_t e = hipSuccess;
        /* entropy decode.  Files with restart markers: on the device, one lane per interval, straight into the
         * device planes (the host only parses headers and finds the markers).  Otherwise, or if that refuses:
         * host threads into pinned memory, then H2D.  Either way chunk k - 1 is on the GPU meanwhile. */
        bool on_device = false;
        if (gpu_entropy) {
            const int grc = ffhip_jpeg_entropy_batch_gpu(files + first, lens + first, cnt, n_threads, &g, sl.d_y, cb ? sl.d_u : nullptr,
                                                         cb ? sl.d_v : nullptr, sl.d_q, status + first, sl.st);
            on_device = grc == FFHIP_OK;
            if (!on_device && grc != FFHIP_EINVAL) { rc = grc; break; }
        }
        if (!on_device) {
            const int erc = ffhip_jpeg_entropy_batch(files + first, lens + first, cnt, n_threads, &g, sl.h_y, cb ? sl.h_u : nullptr,
                                                     cb ? sl.h_v : nullptr, sl.h_q, status + first);
            if (erc && !result) result = erc; /* per-picture codes are in status[]; bad pictures still occupy their place */
            e = hipMemcpyAsync(sl.d_y, sl.h_y, cnt * yb * 2, hipMemcpyHostToDevice, sl.st);
            if (e == hipSuccess && cb) e = hipMemcpyAsync(sl.d_u, sl.h_u, cnt * cb * 2, hipMemcpyHostToDevice, sl.st);
            if (e == hipSuccess && cb) e = hipMemcpyAsync(sl.d_v, sl.h_v, cnt * cb * 2, hipMemcpyHostToDevice, sl.st);
            if (e == hipSuccess) e = hipMemcpyAsync(sl.d_q, sl.h_q, (size_t)cnt * 512, hipMemcpyHostToDevice, sl.st);
            if (e != hipSuccess) { rc = FFHIP_EIO; break; }
        }
        rc = ffhip_jpeg_recon_batch(&g, cnt, sl.d_y, cb ? sl.d_u : nullptr, cb ? sl.d_v : nullptr, sl.d_q, 256, sl.d_out, (int64_t)dev_pitch,
                                    (int64_t)out_b, nullptr, 0, sl.st);
        if (rc) break;
        if (pinned_dst) {
            if (cnt == 1 || image_stride == pitch * height) /* the caller's pictures follow each other row after row: one copy for the chunk */
                e = hipMemcpy2DAsync(bgra + (int64_t)first * image_stride, (size_t)pitch, sl.d_out, dev_pitch, row_b, (size_t)height * cnt, hipMemcpyDeviceToHost, sl.st);
            else
                for (int i = 0; i < cnt && e == hipSuccess; i++)
                    e = hipMemcpy2DAsync(bgra + (int64_t)(first + i) * image_stride, (size_t)pitch, sl.d_out + (size_t)i * out_b, dev_pitch, row_b,
                                         (size_t)height, hipMemcpyDeviceToHost, sl.st);
        } else {
            e = hipMemcpy2DAsync(sl.h_out, row_b, sl.d_out, dev_pitch, row_b, (size_t)height * cnt, hipMemcpyDeviceToHost, sl.st);
        }
        if (e != hipSuccess) { rc = FFHIP_EIO; break; }
        sl.first = first;
        sl.count = cnt;
    }
    for (int s = 0; s < 2; s++) {
        const int r2 = drain(slot[(k + s) & 1]); /* oldest first */
        if (rc == FFHIP_OK) rc = r2;
    }
    return rc ? rc : result;
}

/* Files in, pixels out ON THE DEVICE: for consumers that live on the GPU (a resize, an inference pre-processing
 * stage) nothing but the compressed bytes crosses PCIe.  Coefficient planes are library scratch (kept per stream). */
#define SCRATCH_FILES_DEV 5
extern "C" int ffhip_jpeg_decode_files_device(const uint8_t *const *files, const size_t *lens, int n, int n_threads,
                                              ffhip_jpeg_geom *geom_out, uint8_t *d_bgra, int64_t pitch, int64_t image_stride,
                                              int *status, void *stream)
{
    if (n < 0 || (n > 0 && (!files || !lens || !d_bgra || !status))) return FFHIP_EINVAL;
    if (n == 0) return FFHIP_OK;
    ffhip_jpeg_geom g;
    int w = 0, h = 0;
    int rc = ffhip_jpeg_probe(files[0], lens[0], &g, &w, &h);
    if (rc) return rc;
    if (geom_out) *geom_out = g;
    if (g.mcu_cols <= 0 || g.mcu_rows <= 0 || ffhip_jpeg_workspace_bytes(&g, 1) != 0) return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    const size_t mcus = (size_t)g.mcu_cols * g.mcu_rows;
    const size_t yb = mcus * g.h * g.v * 64, cb = g.ncomp == 3 ? mcus * 64 : 0; /* int16 elements per picture */
    hipStream_t st = (hipStream_t)stream;
    const char *ge = FFHIP_ENV("FFHIP_JPEG_GPU_ENTROPY");
    const char *sy = FFHIP_ENV("FFHIP_JPEG_SYNC"); /* =0: files without restart markers are one lane each on the device, worth it from a thousand files only */
    if (!(ge && ge[0] == '0') && ((ge && ge[0] == '1') || !(sy && sy[0] == '0') || ffhip_jpeg_probe_restart(files[0], lens[0]) > 0 || n >= 1024)) {
        /* entropy decode on the device, a lane per restart interval, straight into planes in library scratch; one reconstruction launch behind it */
        const size_t words = ((size_t)n * (yb + 2 * cb) * 2 + (size_t)n * 512 + 64) / 4 + 16;
        uint8_t *base = (uint8_t *)ffhip_scratch(SCRATCH_FILES_DEV, stream, words);
        if (!base) return FFHIP_ENOMEM;
        int16_t *dy = (int16_t *)base, *du = cb ? dy + (size_t)n * yb : nullptr, *dv = cb ? du + (size_t)n * cb : nullptr;
        uint16_t *dq = (uint16_t *)(base + (((size_t)n * (yb + 2 * cb) * 2 + 15) & ~(size_t)15));
        /* (the reconstruction is enqueued by the entropy call itself, behind each part of the batch as it is decoded) */
        g_ffhip_huff_then.on = 1; g_ffhip_huff_then.bgra = d_bgra; g_ffhip_huff_then.pitch = pitch; g_ffhip_huff_then.image_stride = image_stride;
        rc = ffhip_jpeg_entropy_batch_gpu(files, lens, n, n_threads, &g, dy, du, dv, dq, status, stream);
        g_ffhip_huff_then.on = 0;
        if (rc == FFHIP_OK) return FFHIP_OK;
        if (rc != FFHIP_EINVAL) return rc;
    }
    /* Host threads (files without restart markers are one interval each: a lane per FILE only pays from a thousand files on).  A pipeline of
     * chunks over the two slots ffhip_jpeg_decode_files uses: while the host threads decode chunk k + 1 into pinned memory, chunk k is copied
     * to the device and reconstructed on the slot's own stream, straight into the caller's d_bgra.  (Until round 5 this path decoded the whole
     * batch into pageable vectors, then uploaded it: 1.84 s for 256 4K files, most of it page faults and a pageable copy of 9.5 GB.)  Everything
     * has run when the call returns. */
    int chunk = n_threads < 8 ? 8 : (n_threads > 32 ? 32 : n_threads);
    if (chunk > n) chunk = n;
    FFHIP_CHECK(hipStreamSynchronize(st), FFHIP_EIO); /* d_bgra may still be read by what `stream` holds */
    std::lock_guard<std::mutex> lock(g_pipe_mu);
    Slot *slot = g_slot;
    for (int s = 0; s < 2; s++)
        if (!prepare(slot[s], chunk * yb * 2, chunk * cb * 2, (size_t)chunk * 512, 0, false)) return FFHIP_ENOMEM;
    int result = FFHIP_OK, k = 0;
    rc = FFHIP_OK;
    for (int first = 0; first < n && rc == FFHIP_OK; first += chunk, k++) {
        Slot &sl = slot[k & 1];
        const int cnt = n - first < chunk ? n - first : chunk;
        if (hipStreamSynchronize(sl.st) != hipSuccess) { rc = FFHIP_EIO; break; } /* the slot's previous chunk (k - 2) has left its pinned planes */
        const int erc = ffhip_jpeg_entropy_batch(files + first, lens + first, cnt, n_threads, &g, sl.h_y, cb ? sl.h_u : nullptr, cb ? sl.h_v : nullptr, sl.h_q,
                                                 status + first);
        if (erc && !result) result = erc; /* per-picture codes are in status[]; bad pictures still occupy their place */
        hipError_t e = hipMemcpyAsync(sl.d_y, sl.h_y, cnt * yb * 2, hipMemcpyHostToDevice, sl.st);
        if (e == hipSuccess && cb) e = hipMemcpyAsync(sl.d_u, sl.h_u, cnt * cb * 2, hipMemcpyHostToDevice, sl.st);
        if (e == hipSuccess && cb) e = hipMemcpyAsync(sl.d_v, sl.h_v, cnt * cb * 2, hipMemcpyHostToDevice, sl.st);
        if (e == hipSuccess) e = hipMemcpyAsync(sl.d_q, sl.h_q, (size_t)cnt * 512, hipMemcpyHostToDevice, sl.st);
        if (e != hipSuccess) { rc = FFHIP_EIO; break; }
        rc = ffhip_jpeg_recon_batch(&g, cnt, sl.d_y, cb ? sl.d_u : nullptr, cb ? sl.d_v : nullptr, sl.d_q, 256, d_bgra + (int64_t)first * image_stride, pitch, image_stride,
                                    nullptr, 0, sl.st);
    }
    for (int s = 0; s < 2; s++)
        if (hipStreamSynchronize(slot[s].st) != hipSuccess && rc == FFHIP_OK) rc = FFHIP_EIO;
    return rc ? rc : result;
}
