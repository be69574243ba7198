/*
 * ffhip_runtime.hip -- device binding, memory/stream/event helpers and the copy
 * calibration kernel of libffpic_hip.so.  Plain plumbing around the HIP runtime
 * so that a C11 host (the reference is C) needs no HIP headers.
 */
#include "ffhip_internal.h"

#include <stdio.h>
#include <string.h>

static int g_device = -1;
static int g_ready = 0;
static char g_arch[64] = "";
static char g_last_error[256] = "";

/* FFHIP_* switches: read once per call site and process, re-read after ffhip_reload_env() */
#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
static std::atomic<int> g_env_gen{0};
static std::mutex g_env_mu;
static std::set<std::string> g_env_values; /* interned: element addresses of a std::set never move, nothing is ever erased */
extern "C" const char *ffhip_env_lookup(struct ffhip_env_site *site)
{
    const int gen = g_env_gen.load(std::memory_order_acquire);
    if (__atomic_load_n(&site->gen, __ATOMIC_ACQUIRE) != gen) {
        std::lock_guard<std::mutex> lock(g_env_mu);
        const char *v = getenv(site->name);
        const char *kept = v ? g_env_values.insert(std::string(v)).first->c_str() : nullptr;
        __atomic_store_n(&site->val, kept, __ATOMIC_RELAXED);
        __atomic_store_n(&site->gen, gen, __ATOMIC_RELEASE);
    }
    return __atomic_load_n(&site->val, __ATOMIC_RELAXED);
}
extern "C" void ffhip_reload_env(void) { g_env_gen.fetch_add(1, std::memory_order_acq_rel); }
/* test hook (tests/test_capi.py; needs no device): the value the library holds for switch `name`, by the same lookup a call
 * site makes (a site per name, kept for the process); copies it into dst[0..cap) and returns its full length, or -1 when unset */
extern "C" long ffhip_env_value_test(const char *name, char *dst, size_t cap)
{
    static std::mutex mu;
    static std::map<std::string, ffhip_env_site *> sites;
    if (!name) return -1;
    ffhip_env_site *site;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = sites.find(name);
        if (it == sites.end()) {
            auto ins = sites.emplace(std::string(name), nullptr);
            ins.first->second = new ffhip_env_site{ins.first->first.c_str(), -1, nullptr};
            it = ins.first;
        }
        site = it->second;
    }
    const char *v = ffhip_env_lookup(site);
    if (!v) return -1;
    const size_t n = strlen(v);
    if (dst && cap) { strncpy(dst, v, cap - 1); dst[cap - 1] = 0; }
    return (long)n;
}

extern "C" void ffhip_note_hip_error(int hip_error, const char *what)
{
    snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, hipGetErrorString((hipError_t)hip_error));
    if (FFHIP_ENV("FFHIP_VERBOSE")) fprintf(stderr, "ffpic_hip: %s\n", g_last_error);
}

/* How many single-wave workgroups of `kernel` the device holds at once: what a kernel whose waves WAIT for each other
 * (tickets + progress counters) may launch without a wave holding a ticket it cannot run yet -- and, for two such
 * kernels side by side, what lets both be resident whatever the hardware starts first. */
static std::map<std::tuple<int, const void *, int>, int> g_resident; /* keyed by device too: CU counts differ between parts */
extern "C" int ffhip_resident_waves(const void *kernel, int block_threads)
{
    std::lock_guard<std::mutex> lock(g_env_mu);
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    const auto key = std::make_tuple(dev, kernel, block_threads);
    auto it = g_resident.find(key);
    if (it != g_resident.end()) return it->second;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, 0) != hipSuccess || per_cu < 1 || cus < 1) {
        (void)hipGetLastError();
        return 256; /* one wave per CU of the smallest part: always resident */
    }
    const int n = per_cu * cus;
    g_resident[key] = n;
    return n;
}

extern "C" int ffhip_have_device(void)
{
    if (!g_ready) ffhip_init(g_device < 0 ? 0 : g_device);
    return g_ready;
}

extern "C" int ffhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ffhip_init(int device)
{
    int n = ffhip_device_count();
    if (device < 0 || device >= n) return FFHIP_ENODEV;
    FFHIP_CHECK(hipSetDevice(device), FFHIP_ENODEV);
    hipDeviceProp_t prop;
    FFHIP_CHECK(hipGetDeviceProperties(&prop, device), FFHIP_ENODEV);
    strncpy(g_arch, prop.gcnArchName, sizeof g_arch - 1);
    char *colon = strchr(g_arch, ':');
    if (colon) *colon = 0;
    /* the code object only holds gfx950 ISA: refuse anything else up front */
    if (strcmp(g_arch, "gfx950") != 0) {
        snprintf(g_last_error, sizeof g_last_error, "device %d is %s, this library is built for gfx950 only", device, g_arch);
        return FFHIP_ENODEV;
    }
    g_device = device;
    g_ready = 1;
    return FFHIP_OK;
}

extern "C" void ffhip_release_caches(void); /* below: scratch kept per (stage, stream) */
extern "C" void ffhip_pipeline_release(void); /* ffhip_pipeline.hip: its two slots of pinned + device buffers */
extern "C" void ffhip_vp8_release_side_streams(void); /* ffhip_vp8_lf.hip: the side stream + events of ffhip_vp8_predict_loopfilter */
/* Nothing of the library's may be in flight.  Frees what the library keeps between calls (device scratch, pinned
 * staging, the pipeline's buffers); the next compute call binds the device again. */
extern "C" void ffhip_shutdown(void)
{
    if (g_ready) {
        (void)hipDeviceSynchronize();
        ffhip_pipeline_release();
        ffhip_vp8_release_side_streams();
        ffhip_release_caches();
    }
    g_ready = 0;
}

/* One pinned, device-visible word that a kernel with an in-launch dependency wait writes when it
 * gives up (its bounded spin ran out); checked and cleared by ffhip_stream_sync. */
static int *g_async_err = nullptr;
extern "C" int *ffhip_async_err_word(void)
{
    if (!g_async_err && ffhip_have_device()) {
        if (hipHostMalloc((void **)&g_async_err, 64, hipHostMallocMapped) != hipSuccess) g_async_err = nullptr;
        else *g_async_err = 0;
    }
    return g_async_err;
}

extern "C" const char *ffhip_arch_name(void) { return g_arch; }

/* Device scratch owned by (kind, stream), grown on demand and kept: the dependency-scheduled stages
 * put their schedules / counters there.  Everything a call enqueues is ordered on its stream, so
 * the same stream may reuse its buffer call after call without waiting, and calls on different
 * streams (or threads) never share one.  Growing waits for the stream first: the old buffer may
 * still be read by what that stream has queued. */
#include <utility>
struct ScratchEntry { uint32_t *dev; size_t words; };
static std::map<std::pair<int, void *>, ScratchEntry> g_scratch;
static std::mutex g_scratch_mu;
/* the same for PINNED host memory (staging for uploads): owned by (kind, stream), kept, grown on demand.  The caller
 * must not refill it before what it enqueued from it on that stream has run (a stream sync, as a rule). */
static std::map<std::pair<int, void *>, std::pair<uint8_t *, size_t>> g_pinned;
extern "C" uint8_t *ffhip_pinned_scratch(int kind, void *stream, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    auto &e = g_pinned[std::make_pair(kind, stream)];
    if (bytes > e.second) {
        if (e.first) {
            if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return nullptr;
            (void)hipHostFree(e.first);
        }
        e.first = nullptr;
        e.second = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc((void **)&e.first, want, hipHostMallocDefault) != hipSuccess) { e.first = nullptr; return nullptr; }
        e.second = want;
    }
    return e.first;
}

extern "C" void ffhip_huff_release_thread(void); /* ffhip_huff_gpu.hip: the calling thread's header records */
extern "C" void ffhip_hevc_tiles_release(void); /* ffhip_hevc_intra.hip: the per-stream guards of the tile call's two scratch sets */
extern "C" void ffhip_release_caches(void)
{
    ffhip_huff_release_thread();
    ffhip_hevc_tiles_release();
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto &e : g_scratch)
        if (e.second.dev) (void)hipFree(e.second.dev);
    g_scratch.clear();
    for (auto &e : g_pinned)
        if (e.second.first) (void)hipHostFree(e.second.first);
    g_pinned.clear();
    if (g_async_err) { (void)hipHostFree(g_async_err); g_async_err = nullptr; }
}

extern "C" uint32_t *ffhip_scratch(int kind, void *stream, size_t words)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    ScratchEntry &e = g_scratch[std::make_pair(kind, stream)];
    if (words > e.words) {
        if (e.dev) {
            if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return nullptr;
            (void)hipFree(e.dev);
        }
        e.dev = nullptr;
        e.words = 0;
        const size_t want = words + words / 4 + 1024; /* some headroom: lists of nearly equal size do not reallocate */
        if (hipMalloc((void **)&e.dev, want * sizeof(uint32_t)) != hipSuccess) { e.dev = nullptr; return nullptr; }
        e.words = want;
    }
    return e.dev;
}

extern "C" const char *ffhip_strerror(int code)
{
    switch (code) {
    case FFHIP_OK: return "ok";
    case FFHIP_RETRIED: return "a side-by-side VP8 call was repeated by the sync: its outputs are good, what was enqueued behind it is stale";
    case FFHIP_EINVAL: return "invalid argument or unsupported geometry";
    case FFHIP_ENOMEM: return "out of memory";
    case FFHIP_ENODEV: return g_last_error[0] ? g_last_error : "no usable gfx950 device";
    case FFHIP_EIO: return g_last_error[0] ? g_last_error : "HIP launch or copy failed";
    default: return "unknown error";
    }
}

extern "C" void *ffhip_malloc(size_t bytes)
{
    void *p = nullptr;
    if (!ffhip_have_device()) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    return p;
}
extern "C" void ffhip_free(void *p) { if (p) (void)hipFree(p); }

extern "C" int ffhip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    FFHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), FFHIP_EIO);
    return FFHIP_OK;
}
extern "C" int ffhip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    FFHIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), FFHIP_EIO);
    return FFHIP_OK;
}
extern "C" int ffhip_memset(void *dst, int value, size_t bytes, void *stream)
{
    FFHIP_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream), FFHIP_EIO);
    return FFHIP_OK;
}
extern "C" void *ffhip_stream_create(void)
{
    hipStream_t s = nullptr;
    if (!ffhip_have_device()) return nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return (void *)s;
}
extern "C" void ffhip_stream_destroy(void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
extern "C" int ffhip_vp8_side_by_side_retry(void *stream); /* ffhip_vp8_lf.hip: 0 nothing reported by a side-by-side call of this stream, FFHIP_RETRIED repeated, FFHIP_EIO */
extern "C" void ffhip_vp8_retry_forget(void *stream);
extern "C" int ffhip_stream_sync(void *s)
{
    FFHIP_CHECK(hipStreamSynchronize((hipStream_t)s), FFHIP_EIO);
    /* A bounded wait of a side-by-side VP8 call of THIS stream ran out (reported in a pinned word of that call's own: possible on a device
     * shared with other work, where one of its two kernels may not become resident next to the other).  Its inputs are intact and the one thing
     * of the planes' former contents it reads is kept (the last luma column), so the library repeats the stages ONE AFTER THE OTHER here and
     * says so: FFHIP_RETRIED, not FFHIP_OK -- what the caller had enqueued behind the call has consumed the aborted run's planes. */
    const int retried = ffhip_vp8_side_by_side_retry(s);
    if (retried < 0) {
        snprintf(g_last_error, sizeof g_last_error, "a side-by-side VP8 call aborted and could not be repeated");
        if (g_async_err) *(volatile int *)g_async_err = 0;
        return FFHIP_EIO;
    }
    if (g_async_err && *(volatile int *)g_async_err) {
        const int code = *(volatile int *)g_async_err;
        snprintf(g_last_error, sizeof g_last_error, code == FFHIP_ASYNC_BAD_INPUT ? "a dependency-scheduled kernel refused its input (code %d)"
                                                                                   : "a dependency-scheduled kernel aborted (code %d)", code);
        *(volatile int *)g_async_err = 0;
        return code == FFHIP_ASYNC_BAD_INPUT ? FFHIP_EINVAL : FFHIP_EIO;
    }
    ffhip_vp8_retry_forget(s);
    return retried;
}
extern "C" void *ffhip_event_create(void)
{
    hipEvent_t e = nullptr;
    if (!ffhip_have_device()) return nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void *)e;
}
extern "C" void ffhip_event_destroy(void *e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
extern "C" int ffhip_event_record(void *e, void *s)
{
    FFHIP_CHECK(hipEventRecord((hipEvent_t)e, (hipStream_t)s), FFHIP_EIO);
    return FFHIP_OK;
}
extern "C" float ffhip_event_elapsed_ms(void *start, void *stop)
{
    float ms = -1.0f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.0f;
    return ms;
}

/* 16 B per lane non-temporal copy, one element per thread (no loop): the launch shape and
 * cache policy that stream fastest on MI355X (tests/tools/membench.hip).  The achievable-HBM yardstick the fused
 * kernel is compared with in the same process (bench.py). */
__global__ __launch_bounds__(256) void k_copy_calibrate(u32x4 *dst, const u32x4 *src, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}

extern "C" int ffhip_copy_calibrate(void *d_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!d_dst || !d_src || (bytes & 15) || ((uintptr_t)d_dst & 15) || ((uintptr_t)d_src & 15)) return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    size_t n16 = bytes / 16;
    if (!n16) return FFHIP_OK;
    if (n16 > (size_t)0x7fffffff * 256) return FFHIP_EINVAL;
    hipLaunchKernelGGL(k_copy_calibrate, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (u32x4 *)d_dst, (const u32x4 *)d_src, n16);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
