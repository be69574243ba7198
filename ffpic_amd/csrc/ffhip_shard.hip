/*
 * ffhip_shard.hip -- batches of independent images over the GPUs of one node, from C.
 *
 * The hot path has no exchange step: images are independent, so a batch is cut into contiguous image ranges, one per
 * GPU, every rank runs the same single-GPU launches on its range, and the batch is closed by ONE RCCL all-gather of a
 * 32-byte {rank, status, first, count, checksum} record per rank, which doubles as the batch barrier.  The reference
 * has no counterpart (no NCCL/MPI/threads anywhere); its single-image decode loop (format/jpg.c:458-585)
 * is what each rank's range replaces.
 *
 * RCCL is bound at run time (dlopen), so that libffpic_hip.so loads on a host without it and a one-GPU caller never
 * touches it: the copy already in the process (a host that linked RCCL, or PyTorch's) is preferred over a second one.
 * One process per GPU: rank 0 makes an id (ffhip_comm_unique_id), the host program gets its 128 bytes to the other
 * ranks by whatever means it has (a file, a socket, torch.distributed in bench.py), every rank calls
 * ffhip_comm_init_rank on its own device.
 */
#include "ffhip_internal.h"

#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
/* a build host without the RCCL headers: the five entry points this file binds with dlsym, as rccl.h declares them (the
 * library is optional at run time, so its header is optional at build time; without librccl the comm functions return
 * FFHIP_ENODEV / NULL) */
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclAllGather(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
const char *ncclGetErrorString(ncclResult_t);
}
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    static const char *const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    /* a copy that is already mapped (matched by its soname) first: two RCCLs in one process would each open the devices */
    for (const char *n : names)
        if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (const char *forced = FFHIP_ENV("FFHIP_RCCL_LIB")) {
        if (!g_rccl.lib) g_rccl.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    }
    for (const char *n : names)
        if (!g_rccl.lib) g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.lib) return;
    g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.all_gather = (decltype(g_rccl.all_gather))dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.error_string = (decltype(g_rccl.error_string))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_gather;
}
bool have_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.ok;
}
void note_rccl(ncclResult_t r, const char *what)
{
    char msg[200];
    snprintf(msg, sizeof msg, "%s: rccl: %s", what, g_rccl.error_string ? g_rccl.error_string(r) : "error");
    ffhip_note_hip_error((int)hipErrorUnknown, msg);
}

struct Comm {
    ncclComm_t comm;
    int rank, world;
    ffhip_batch_record *d_send, *d_recv; /* device: one record, world records */
    ffhip_batch_record *h_recv;          /* pinned: world records, then this rank's outgoing record */
};

/* per image: sum over its 32-bit pixel words w[i] of w[i] * ((i & 0xffff) + 1), modulo 2^64 -- position-sensitive within
 * a row pair, cheap, and a plain sum over rows, so that it can be formed in any order */
__global__ __launch_bounds__(256) void k_bgra_checksum(const uint8_t *bgra, long long pitch, long long image_stride, int width,
                                                       int height, unsigned long long *sums)
{
    const int img = blockIdx.y;
    const uint32_t *base = (const uint32_t *)(bgra + (long long)img * image_stride);
    unsigned long long acc = 0;
    const long long words = (long long)width * height;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / width), x = (int)(i - (long long)y * width);
        const uint32_t w = *(const uint32_t *)((const uint8_t *)base + (long long)y * pitch + 4ll * x);
        acc += (unsigned long long)w * (unsigned long long)((i & 0xffff) + 1);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[img], part[0] + part[1] + part[2] + part[3]);
}
} // namespace

/* contiguous range of the images rank owns; sizes differ by at most one */
extern "C" int ffhip_shard_range(long long n_images, int rank, int world, long long *first, long long *count)
{
    if (!first || !count || world < 1 || rank < 0 || rank >= world || n_images < 0) return FFHIP_EINVAL;
    const long long base = n_images / world, extra = n_images % world;
    *first = rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
    return FFHIP_OK;
}

/* 1 when the records tile [0, n_images) exactly, every rank appears once and every status is 0 */
extern "C" int ffhip_batch_complete(const ffhip_batch_record *records, int world, long long n_images)
{
    if (!records || world < 1 || n_images < 0) return 0;
    long long total = 0;
    for (int r = 0; r < world; r++) {
        if (records[r].status != 0 || records[r].count < 0 || records[r].first < 0 || records[r].first + records[r].count > n_images) return 0;
        if (records[r].rank != r) return 0;
        total += records[r].count;
    }
    /* ranges that are inside [0, n), carry n images in total and never overlap tile it */
    for (int a = 0; a < world; a++)
        for (int b = a + 1; b < world; b++) {
            const ffhip_batch_record &p = records[a], &q = records[b];
            if (p.count > 0 && q.count > 0 && p.first < q.first + q.count && q.first < p.first + p.count) return 0;
        }
    return total == n_images;
}

extern "C" int ffhip_bgra_checksum(const uint8_t *d_bgra, int64_t pitch, int64_t image_stride, int width, int height,
                                   int n_images, uint64_t *d_sums, void *stream)
{
    if (n_images < 0 || width <= 0 || height <= 0 || pitch < (int64_t)width * 4 || (pitch & 3) || (image_stride & 3) ||
        ((uintptr_t)d_bgra & 3) || (n_images > 0 && (!d_bgra || !d_sums))) return FFHIP_EINVAL;
    if (n_images == 0) return FFHIP_OK;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    hipStream_t st = (hipStream_t)stream;
    FFHIP_CHECK(hipMemsetAsync(d_sums, 0, (size_t)n_images * 8, st), FFHIP_EIO);
    const long long words = (long long)width * height;
    int bx = (int)((words + 256 * 16 - 1) / (256 * 16));
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_bgra_checksum, dim3(bx, n_images), dim3(256), 0, st, d_bgra, (long long)pitch, (long long)image_stride, width, height,
                       (unsigned long long *)d_sums);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}

extern "C" int ffhip_comm_unique_id(void *id128)
{
    if (!id128) return FFHIP_EINVAL;
    if (!have_rccl()) return FFHIP_ENODEV;
    static_assert(sizeof(ncclUniqueId) == FFHIP_COMM_ID_BYTES, "the id travels as 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.get_unique_id(&id);
    if (r != ncclSuccess) { note_rccl(r, "ncclGetUniqueId"); return FFHIP_EIO; }
    memcpy(id128, &id, sizeof id);
    return FFHIP_OK;
}

extern "C" void *ffhip_comm_init_rank(const void *id128, int rank, int world)
{
    if (!id128 || world < 1 || rank < 0 || rank >= world) return nullptr;
    if (!ffhip_have_device() || !have_rccl()) return nullptr;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    Comm *c = (Comm *)calloc(1, sizeof *c);
    if (!c) return nullptr;
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = g_rccl.comm_init_rank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { note_rccl(r, "ncclCommInitRank"); free(c); return nullptr; }
    if (hipMalloc((void **)&c->d_send, sizeof(ffhip_batch_record)) != hipSuccess ||
        hipMalloc((void **)&c->d_recv, sizeof(ffhip_batch_record) * (size_t)world) != hipSuccess ||
        hipHostMalloc((void **)&c->h_recv, sizeof(ffhip_batch_record) * ((size_t)world + 1), hipHostMallocDefault) != hipSuccess) {
        ffhip_comm_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void ffhip_comm_destroy(void *comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return;
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->comm && g_rccl.ok) (void)g_rccl.comm_destroy(c->comm);
    free(c);
}

/* Closes a batch: this rank's record goes out, everybody's come back (ncclAllGather of 32 bytes per rank on `stream`,
 * behind whatever the rank has queued there), `stream` is synchronised and h_records[world] filled in rank order.
 * comm == NULL is the one-GPU case: the stream is synchronised and h_records[0] is this rank's record. */
extern "C" int ffhip_batch_close(void *comm, int rank, int world, long long first, long long count, int status, uint64_t checksum,
                                 ffhip_batch_record *h_records, void *stream)
{
    if (!h_records || world < 1 || rank < 0 || rank >= world || count < 0) return FFHIP_EINVAL;
    ffhip_batch_record mine;
    memset(&mine, 0, sizeof mine);
    mine.rank = rank;
    mine.status = status;
    mine.first = first;
    mine.count = count;
    mine.checksum = checksum;
    hipStream_t st = (hipStream_t)stream;
    Comm *c = (Comm *)comm;
    if (!c) {
        if (world != 1) return FFHIP_EINVAL;
        if (ffhip_have_device()) {
            const int rc = ffhip_stream_sync(stream); /* turns an in-launch abort into FFHIP_EIO as well */
            if (rc) return rc;
        }
        h_records[0] = mine;
        return FFHIP_OK;
    }
    /* every check that can fail on ONE rank sits in front of the collective: a rank that left early would leave the
     * others in the gather.  From here on a failure is the device's (or RCCL's), which every rank sees. */
    if (c->world != world || c->rank != rank) return FFHIP_EINVAL;
    ffhip_batch_record *h_send = c->h_recv + world; /* pinned, lives as long as the communicator: the async copy may read it late */
    *h_send = mine;
    FFHIP_CHECK(hipMemcpyAsync(c->d_send, h_send, sizeof mine, hipMemcpyHostToDevice, st), FFHIP_EIO);
    const ncclResult_t r = g_rccl.all_gather(c->d_send, c->d_recv, sizeof mine, ncclChar, c->comm, st);
    if (r != ncclSuccess) { note_rccl(r, "ncclAllGather"); return FFHIP_EIO; }
    FFHIP_CHECK(hipMemcpyAsync(c->h_recv, c->d_recv, sizeof mine * (size_t)world, hipMemcpyDeviceToHost, st), FFHIP_EIO);
    const int rc = ffhip_stream_sync(stream);
    if (rc) return rc;
    memcpy(h_records, c->h_recv, sizeof mine * (size_t)world);
    return FFHIP_OK;
}
