/* ffhip_internal.h -- shared by the .hip translation units of libffpic_hip.so. */
#ifndef FFHIP_INTERNAL_H
#define FFHIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ffpic_hip.h"

#define FFHIP_CHECK(expr, code)                    \
    do {                                           \
        hipError_t e__ = (expr);                   \
        if (e__ != hipSuccess) {                   \
            ffhip_note_hip_error((int)e__, #expr); \
            return (code);                         \
        }                                          \
    } while (0)

#ifdef __cplusplus
extern "C" {
#endif
void ffhip_note_hip_error(int hip_error, const char *what);
int ffhip_have_device(void); /* 1 once ffhip_init succeeded on a gfx950 device */
uint32_t *ffhip_scratch(int kind, void *stream, size_t words);
uint8_t *ffhip_pinned_scratch(int kind, void *stream, size_t bytes); /* per (kind, stream) pinned host staging, NULL on failure */ /* per (kind, stream) device scratch, NULL on failure */
int *ffhip_async_err_word(void); /* pinned word kernels report an in-launch abort through; see ffhip_stream_sync */
/* The FFHIP_* switches (A/B knobs of the tools, diagnostics) are read from the environment ONCE per process, at a call
 * site's first use, and kept; ffhip_reload_env() (public, include/ffpic_hip.h) makes every site read its switch again. */
/* `val` points into storage the library keeps for the life of the process (values are interned, whatever their length:
 * FFHIP_RCCL_LIB is a path), so a pointer a caller got stays valid and unchanged across ffhip_reload_env(); `gen` and `val`
 * are only touched with atomic loads / stores (val first, then gen with release), so lookups may race with a reload. */
struct ffhip_env_site { const char *name; int gen; const char *val; };
const char *ffhip_env_lookup(struct ffhip_env_site *site); /* NULL when unset */
int ffhip_resident_waves(const void *kernel, int block_threads); /* workgroups of `block_threads` threads of `kernel` the device holds at once (occupancy x CUs), cached */
#ifdef __cplusplus
}
#endif

/* codes a dependency-scheduled kernel leaves in the async error word: 1-3 a bounded wait ran out (FFHIP_EIO), 4 the kernel
 * met input it refuses, e.g. a mode byte no VP8 stream can hold (FFHIP_EINVAL) */
#define FFHIP_ASYNC_BAD_INPUT 4

#ifdef __cplusplus
#define FFHIP_ENV(NAME) ([]() -> const char * { static struct ffhip_env_site site = {NAME, -1, nullptr}; return ffhip_env_lookup(&site); }())
#endif

typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifdef __HIPCC__
/* Device-coherent ("sc1") loads that are ORDINARY loads to the compiler: a relaxed agent-scope
 * __hip_atomic_load of a sub-dword type gets an s_waitcnt vmcnt(0) right behind it (its extension
 * is a separate instruction), which serialises every fetch; a raw buffer load with the sc1 bit in
 * its cache policy is the same memory operation and is waited for at first use only.
 * Out-of-range offsets (offset >= bytes, so negative ones too) read as 0. */
#define FFHIP_AUX_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ffhip_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int ffhip_load_u8_sc1(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return (int)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, byte_off, 0, FFHIP_AUX_SC1);
}
__device__ __forceinline__ int ffhip_load_s16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return (int)(short)__builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, FFHIP_AUX_SC1);
}
#endif


/* ffhip_vp8_predict_loopfilter: the two row kernels of one call side by side (ffhip_vp8_lf.hip).  While `active`, the
 * prediction entry records `fork` behind its counter reset and publishes where its per-row counters live, and the loop-filter
 * entry launches on `side` behind `fork`, polling those counters. */
struct FfhipVp8Fusion {
    int active;
    const uint32_t *pred_progress;
    void *side;  /* hipStream_t */
    void *fork;  /* hipEvent_t  */
    int pred_split; /* the prediction runs luma and chroma rows apart: its chroma counters (behind the luma ones) count too */
    int pshift; /* a row's progress counter is word (image * mbrows + row) << pshift */
    int *err_word; /* where the two kernels of THIS call report a bounded wait that ran out: the call's own pinned word (its retry record's), so that
                      nobody else's abort can set off the retry; NULL = the process-wide word */
};
extern thread_local FfhipVp8Fusion g_ffhip_vp8_fusion;
/* ffhip_vp8_decode_frames (row form) -> ffhip_vp8_predict_loopfilter: the colour conversion the caller enqueues behind the call belongs to it -- a
 * retry has to run it again, behind the filter (consumed, i.e. switched off, by the callee) */
struct FfhipVp8Then { int on; uint8_t *bgra; int pitch; int64_t image_stride; };
extern thread_local FfhipVp8Then g_ffhip_vp8_then;

/* ffhip_jpeg_decode_files_device -> ffhip_jpeg_entropy_batch_gpu: the reconstruction of the pictures, enqueued by the entropy call itself behind each
 * part of the batch it has decoded (consumed, i.e. switched off, by the callee) */
struct FfhipHuffThen { int on; uint8_t *bgra; int64_t pitch, image_stride; };
extern thread_local FfhipHuffThen g_ffhip_huff_then;

/* the calling thread's side stream with its fork / join events (ffhip_vp8_lf.hip: one set per thread and device, released by ffhip_shutdown) */
struct FfhipSide { void *stream, *fork, *join, *mid, *aux; }; /* mid: a second point of the main stream the side stream may wait for; aux: a second
                                                                point of the side stream the main stream may wait for */
extern "C" int ffhip_side_stream_get(FfhipSide *out);
/* the calling thread's streams and events for a pipelined call (ffhip_hevc_intra_recon_tiles): a stream the chunks' pre-passes follow each other
 * on, a second stream for grouped kernels (the caller's is the first), events; same owner and lifetime as the side stream */
#define FFHIP_PIPE_EVENTS 12
struct FfhipPipe { void *plan, *groups2, *ev[FFHIP_PIPE_EVENTS]; };
extern "C" int ffhip_pipe_streams_get(FfhipPipe *out);

/* ... and the device Huffman decoder's (ffhip_jpeg_entropy_batch_gpu): the copy stream its parts' bytes go up on, the second kernel stream, the
 * parts' events, fork / join, two timing events; all made together (none is published unless all exist), keyed by the thread's device like
 * the side stream and released with it */
#define FFHIP_HUFF_PARTS 8
struct FfhipHuffStreams { void *up, *c2, *part_ev[FFHIP_HUFF_PARTS], *fork, *join, *time_ev[2]; };
extern "C" int ffhip_huff_streams_get(FfhipHuffStreams *out);

/* bits of a schedule slot's program word (second quarter, .x) that two files know: k_hevc_intra_program writes the word, k_plan_emit adds
 * what only the planner knows when the programs were built NEXT TO it (ffhip_hevc_intra.hip has the rest of the layout) */
#define FFHIP_PK_KIND_MASK 7u
#define FFHIP_PK_SIGNAL 64u
#define FFHIP_PK_WAIT 128u
#define FFHIP_PK_SLOW 256u
#define FFHIP_PROG_NO_RESIDUAL 0xffffff00u

/* Control words of the VP8 row kernels (ffhip_vp8_pred.hip, ffhip_vp8_lf.hip; the scratch they live in starts on a 256-byte boundary): the
 * ticket counters -- one device-scope atomic per row from every wave -- and the abort word -- read by waiting waves between polls -- each
 * ALONE in a 128-byte line, the per-row progress counters behind them.  (Until late in round 4 they were words 0, 2 and 1 of one line, with
 * the first progress counters behind them in the same line: every poll of a waiting wave queued up with the ticket atomics.) */
#define FFHIP_VP8_CTRL_TICKET_C 32
#define FFHIP_VP8_CTRL_ABORT 64
#define FFHIP_VP8_CTRL_HDR 128
#define FFHIP_VP8_LF_CTRL_ABORT 32
#define FFHIP_VP8_LF_CTRL_HDR 64

/* What ffhip_hevc_intra_recon hangs into the device planner's chain of launches (ffhip_hevc_plan_gpu_checked): everything is optional.
 *   after_check       behind the list's validation kernel: start what depends on the TU list alone (the substitution table, side stream)
 *   ticket_stream     behind k_plan_owner: a stream, already waiting for that kernel, that takes the depth sweep NOW -- next to k_plan_count
 *                     -- and the ticket kernels and k_plan_emit later (NULL: everything in line on the planner's stream)
 *   after_count       behind k_plan_count: flags and wait counts are final (the per-pixel programs)
 *   tickets_wait      (ticket_stream given) make that stream wait for k_plan_count, in front of the ticket kernels
 *   tickets_enqueued  (ticket_stream given) the schedule's last kernel is on that stream: whoever reads the schedule waits for it
 *   by_plane          the list interleaves the planes inside a scheduling window (the reference's own order, coding/hevc.c:5013-5180): plan and
 *                     decode a stable partition of it by plane; *tus_used then points at that copy (in the planner's scratch) */
struct FfhipPlanHooks {
    void *ctx;
    int (*after_check)(void *ctx, const unsigned *refused);
    void *(*ticket_stream)(void *ctx);
    int (*after_count)(void *ctx, const unsigned char *flags, const unsigned *wcount, const unsigned *result);
    int (*tickets_wait)(void *ctx);
    int (*tickets_enqueued)(void *ctx);
    int by_plane;                       /* sort the list by plane in front of k_plan_owner (ffhip_hevc_plan_gpu.hip, k_part_*) */
    const ffhip_hevc_tu **tus_used;     /* out, set before after_count runs: the records the schedule's TU indices refer to (the caller's, or the sorted copy) */
};

#endif
