/*
 * ffpic_hip.h -- C ABI of libffpic_hip.so, the MI355X (gfx950) back-end for the
 * post-entropy reconstruction stage of the ffpic image decoder.
 *
 * Everything here is plain C: pointers, sizes, ints.  No C++/torch types.
 * Three groups of entry points, each citing the reference interface it stands
 * behind (paths relative to the ffpic source tree):
 *
 *  (1) the accelerator registry seam           arch/accl.h:13-35, arch/accl.c:17-62
 *  (2) the built-in op tables JPEG decodes by  utils/idct.h:14-25, utils/colorspace.h:29-35
 *  (3) a batched, device-resident extension    (new; the per-block ABI of (1)/(2)
 *      cannot be fast on a GPU -- arch/opencl/opcl.c:42-88 shows why)
 *
 * Error convention of (3): 0 on success, negative errno-style code otherwise
 * (FFHIP_E*).  (1) and (2) return void like the reference; a back-end that
 * cannot run does not register (arch/opencl/opcl.c:112-114), callers fall back
 * to C when the lookup returns NULL (format/webp.c:1173, coding/hevc.c:3913-3919).
 */
#ifndef FFPIC_HIP_H
#define FFPIC_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFHIP_ABI_VERSION 1

/* error codes (negative errno values) */
#define FFHIP_OK 0
#define FFHIP_EINVAL (-22)  /* bad argument / unsupported geometry          */
#define FFHIP_ENOMEM (-12)  /* device or host allocation failed             */
#define FFHIP_ENODEV (-19)  /* no usable gfx950 device / HIP runtime error  */
#define FFHIP_EIO    (-5)   /* kernel launch or copy failed                 */
#define FFHIP_RETRIED 1     /* ffhip_stream_sync only, not an error: a side-by-side VP8 call on the stream ran into a bounded wait and was
                               repeated by the sync; its outputs are now those of an undisturbed call, but whatever the CALLER had enqueued
                               behind it on the stream has consumed the aborted run's output and must be enqueued again */

/* ------------------------------------------------------------------ runtime */

/* Number of visible HIP devices (0 when there is no GPU; never fails). */
int ffhip_device_count(void);
/* Bind the calling thread's library state to `device` (hipSetDevice) and create
 * the small internal staging buffers used by the per-block entry points. */
int ffhip_init(int device);
void ffhip_shutdown(void); /* with nothing in flight: frees the scratch, staging and pipeline buffers the library keeps
                              between calls; a later compute call binds the device again */
const char *ffhip_strerror(int code);
/* The FFHIP_* environment switches (A/B knobs of tests/tools, diagnostics; none is needed in production) are read ONCE per
 * process, at first use.  A host that changes one in a live process calls this to have them read again. */
void ffhip_reload_env(void);
/* Test hook (no device needed): what the library holds for the switch `name` -- copied into dst[0..cap), full length returned,
 * -1 when unset.  Values are kept whole whatever their length (FFHIP_RCCL_LIB is a path). */
long ffhip_env_value_test(const char *name, char *dst, size_t cap);
/* "gfx950" etc. of the bound device, "" if none. */
const char *ffhip_arch_name(void);

/* Device memory / stream / event helpers so that a pure-C host (the reference is
 * C11) can drive the batched API without linking the HIP runtime itself. */
void *ffhip_malloc(size_t bytes);
void ffhip_free(void *dptr);
int ffhip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int ffhip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int ffhip_memset(void *dst, int value, size_t bytes, void *stream);
void *ffhip_stream_create(void);
void ffhip_stream_destroy(void *stream);
int ffhip_stream_sync(void *stream); /* NULL = the default stream; FFHIP_EIO also if a dependency-scheduled
                                         kernel (VP8 predict / loop filter, HEVC intra) reported an abort;
                                         FFHIP_RETRIED (> 0) when a side-by-side VP8 call was repeated, see there */
void *ffhip_event_create(void);
void ffhip_event_destroy(void *event);
int ffhip_event_record(void *event, void *stream);
/* milliseconds between two recorded events (synchronises on `stop`); <0 on error */
float ffhip_event_elapsed_ms(void *start, void *stop);

/* ------------------------------------------- (1) accelerator registry seam */

/* New member of `enum simd_type` (arch/accl.h:13-18 uses 1, 2, 25, 26). */
#define GPU_TYPE_HIP 27

/* Layout-compatible with `struct accl_ops` (arch/accl.h:20-25) on LP64:
 * fn ptrs @0,@8; type @16; TAILQ_ENTRY{tqe_next @24, tqe_prev @32}; sizeof 40. */
struct ffhip_accl_ops {
    void (*idct_4x4)(int16_t *in, int bitdepth); /* VP8 4x4, == idct_4x4_16 (utils/idct.c:100-151) */
    void (*idct_8x8)(int16_t *in, int bitdepth); /* JPEG 8x8, == idct_8x8_16 (utils/idct.c:512-534) */
    int type;                                    /* GPU_TYPE_HIP */
    struct {
        struct ffhip_accl_ops *tqe_next;
        struct ffhip_accl_ops **tqe_prev;
    } next;
};

/* Same protocol as x86_sse2_init / opcl_amd_init / vulkan_init (arch/accl.c:21-35):
 * on success registers the static ops with accl_ops_register() when that symbol
 * exists in the process (i.e. libffpic is loaded); registers nothing when no
 * gfx950 device can be initialised.  hip_accl_uninit mirrors opcl_amd_uninit. */
void hip_accl_init(void);
void hip_accl_uninit(void);
/* The ops struct itself, or NULL when no device could be initialised. */
struct ffhip_accl_ops *ffhip_accl_ops_get(void);

/* ------------------------------------------------ (2) built-in op tables   */

/* Layout-compatible with `struct dct_ops` (utils/idct.h:14-21). */
struct ffhip_dct_ops {
    int bitdepth;
    void (*idct_4x4)(void *in, int bitdepth);
    void (*idct_8x8)(void *in, int bitdepth);
    void (*fdct_4x4)(void *in); /* NULL: encoder side, out of scope */
    void (*fdct_8x8)(void *in); /* NULL */
};
/* Layout-compatible with `struct cs_ops` (utils/colorspace.h:29-33). */
struct ffhip_cs_ops {
    void (*YUV_to_BGRA32)(uint8_t *dst, int pitch, void *Y, void *U, void *V, int vertical,
                          int horizontal);
    void (*YUV420_to_BGRA32)(uint8_t *dst, int pitch, void *Y, void *U, void *V); /* NULL as in the reference */
};
/* Replacements for get_dct_ops(16) / get_cs_ops(16) (utils/idct.c:829-832,
 * utils/colorspace.c:788-791).  Only the 16-bit tables exist (that is what
 * format/jpg.c:467-468 asks for); NULL for other depths or without a device. */
const struct ffhip_dct_ops *ffhip_get_dct_ops(int component_bits);
const struct ffhip_cs_ops *ffhip_get_cs_ops(int component_bits);
/* HEVC DST-VII 4x4, same signature and rounding as idct_4x4_hevc (utils/idct.h:25,
 * utils/idct.c:36-55).  Kept as its own entry: the reference back-ends overloaded
 * accl_ops.idct_4x4 with three different transforms. */
void ffhip_idct_4x4_hevc(const int16_t *in, int16_t *out, int bitdepth, bool epp);

/* ------------------------------------------ (3) batched device-resident API */

/* One batch = n_images pictures of identical geometry.  Coefficient planes are
 * what the entropy decoder of format/jpg.c produces per data unit
 * (decode_data_unit, jpg.c:521-539): quantised, natural order, int16, 64 per
 * block; blocks of a component are in MCU order,
 *     block index = mcu * (h_c * v_c) + vi * h_c + hi,   mcu = my * mcu_cols + mx
 * the per-MCU scratch order the reference feeds to idct_8x8 / YUV_to_BGRA32
 * (jpg.c:545-547, colorspace.c:148).  Chroma is one block per MCU (the
 * reference's colour converter supports nothing else, colorspace.c:149-150).
 * Image i starts at plane + i * blocks_per_image_c * 64. */
typedef struct ffhip_jpeg_geom {
    int32_t mcu_cols, mcu_rows; /* MCUs per row / column                     */
    int32_t ncomp;              /* 1 (grey: U = V = zeros, jpg.c:501,552) or 3 */
    int32_t h, v;               /* luma sampling factors, h*v <= 4 data units per MCU: what the reference's MCU
                                   scratch Y[3][64*4] holds (jpg.c:501) and YUV_to_BGRA32_16bit (colorspace.c:143-150)
                                   converts -- 1x1, 2x1, 1x2, 2x2, 4x1 (4:1:1), 1x4, 3x1, 1x3 */
    int32_t qt_id[3];           /* DQT slot per component, 0..3               */
} ffhip_jpeg_geom;

/* dequant (jpg.c:247-253) + idct_8x8_16 (idct.c:512-534) + YUV_to_BGRA32_16bit
 * (colorspace.c:133-172) for a whole batch.  All pointers are DEVICE pointers.
 *   d_quant       uint16 [4][64] natural order per image (jpg.h:121-130 `dqt.tdata`),
 *                 image i at d_quant + i*quant_stride (elements); stride 0 = shared
 *   d_bgra        B,G,R,0xFF bytes; pixel (x,y) of image i at
 *                 d_bgra + i*image_stride + y*pitch + 4*x ; coded size is
 *                 (8*h*mcu_cols) x (8*v*mcu_rows); pitch >= 4*width, multiple of 16
 *   d_workspace   scratch of ffhip_jpeg_workspace_bytes(geom, n) bytes: 0 (pass NULL) for every
 *                 layout an encoder writes -- 4:2:0, 4:4:4, 4:2:2, 4:4:0, 4:1:1 (h = 4), its
 *                 transpose (v = 4) and grey run fully fused -- and non-zero only for one component
 *                 with h*v > 1 blocks per MCU and for the three-block pairs (h or v = 3)
 *   stream        hipStream_t (NULL = default stream); the call only enqueues. */
int ffhip_jpeg_recon_batch(const ffhip_jpeg_geom *geom, int n_images, const int16_t *d_coef_y,
                           const int16_t *d_coef_u, const int16_t *d_coef_v,
                           const uint16_t *d_quant, int64_t quant_stride, uint8_t *d_bgra,
                           int64_t pitch, int64_t image_stride, void *d_workspace,
                           size_t workspace_bytes, void *stream);
size_t ffhip_jpeg_workspace_bytes(const ffhip_jpeg_geom *geom, int n_images);
/* The BGRA layout this library recommends to a caller that owns its output buffer: *pitch = the reference's row pitch
 * (4 bytes x the coded width, format/jpg.c:484-486) + 1024 bytes, *image_stride = pitch x coded height.  The fused kernels
 * write 16 rows of a macroblock row at once, and with rows exactly 15 360 bytes apart (a 3840-pixel row) the rate depends on
 * where the buffer landed in physical memory (5.7-5.9 or 6.2-6.6 TB/s: DESIGN.md 5); a kibibyte more per row is the one
 * pitch that never measured slower and recovers 0.15-0.26 TB/s on the slow placements.  The drop-in path keeps the
 * reference's pitch, and every entry point takes whatever pitch the caller passes. */
int ffhip_bgra_layout(const ffhip_jpeg_geom *geom, int64_t *pitch, int64_t *image_stride);

/* Same computation from HOST buffers (copies in, runs, copies out, synchronises): what a patched
 * format/jpg.c would call per picture.  The device staging is library scratch kept between calls
 * (grown on demand, released by ffhip_shutdown); calls from several host threads are serialised
 * inside the library (one picture at a time, like the reference's decode loop) -- a host that wants
 * pictures in flight side by side uses the device-pointer entry above with its own buffers and streams. */
int ffhip_jpeg_recon_batch_host(const ffhip_jpeg_geom *geom, int n_images, const int16_t *coef_y,
                                const int16_t *coef_u, const int16_t *coef_v,
                                const uint16_t *quant, int64_t quant_stride, uint8_t *bgra,
                                int64_t pitch, int64_t image_stride);

/* Name and timing of the dominant kernel of the last ffhip_jpeg_recon_batch call
 * geometry class, for bench.py's roofline object. */
const char *ffhip_jpeg_kernel_name(const ffhip_jpeg_geom *geom);

/* ---- planar YUV -> BGRA (WebP frame / HEVC picture colour conversion) ----
 * Same arguments as the reference functions, plus a batch dimension; all pointers are
 * DEVICE pointers, strides in samples (planes) or bytes (pitch, image_stride).
 *   ffhip_yuv420_to_bgra     == YUV420_to_BGRA32        (utils/colorspace.c:291-329; format/webp.c:1868)
 *   ffhip_yuv420_to_bgra_16  == YUV420_to_BGRA32_16bit  (utils/colorspace.c:628-669; coding/hevc.c:7260-7270)
 *   ffhip_yuv400_to_bgra_16  == YUV400_to_BGRA32_16bit  (utils/colorspace.c:715-742; coding/hevc.c:7271-7277)
 * Image i reads planes at +i*plane_stride_* and writes at d_bgra + i*image_stride. */
int ffhip_yuv420_to_bgra(uint8_t *d_bgra, int pitch, const uint8_t *d_y, const uint8_t *d_u,
                         const uint8_t *d_v, int y_stride, int uv_stride, int mbrows, int mbcols,
                         int n_images, int64_t plane_stride_y, int64_t plane_stride_uv,
                         int64_t image_stride, void *stream);
int ffhip_yuv420_to_bgra_16(uint8_t *d_bgra, int pitch, const int16_t *d_y, const int16_t *d_u,
                            const int16_t *d_v, int y_stride, int uv_stride, int ctbrows, int ctbcols,
                            int ctbsize, int n_images, int64_t plane_stride_y,
                            int64_t plane_stride_uv, int64_t image_stride, void *stream);
int ffhip_yuv400_to_bgra_16(uint8_t *d_bgra, int pitch, const int16_t *d_y, int y_stride,
                            int ctbrows, int ctbcols, int ctbsize, int n_images,
                            int64_t plane_stride_y, int64_t image_stride, void *stream);

/* ---- HEIF image grid ----
 * ffhip_heif_grid_parse reads the ImageGrid item payload exactly as decode_grid_items does
 * (format/heif.c:273-298: version, flags, rows_minus_one, columns_minus_one, then 16- or 32-bit
 * big-endian output_width/height by flags & 1); host only.
 * ffhip_heif_grid_compose places rows*cols decoded BGRA tiles (tile j of the row-major `dimg`
 * list at d_tiles + j*tile_stride, tile_pitch bytes per row, all tile_w x tile_h) on the canvas
 * at (j % cols * tile_w, j / cols * tile_h), cropped to out_w x out_h.  NEW behaviour: the
 * reference decodes every tile into the same buffer (heif.c:305) and never places them. */
/* The picture buffer the reference's HEVC decoder allocates per slice and hands to the colour converter
 * (coding/hevc.c:7223-7236, 7258-7277): one int16 buffer of 2*size samples, Y at 0, Cb at `u_offset`, Cr at
 * `v_offset`; what ffhip_hevc_intra_recon / ffhip_yuv420_to_bgra_16 take as their plane pointers, strides and
 * ctb counts.  Host only. */
typedef struct ffhip_hevc_layout {
    int32_t height;              /* pic_height_in_luma_samples rounded up to 4                      */
    int32_t y_stride, uv_stride; /* width rounded up to 4; half of it                               */
    int64_t size;                /* height * y_stride: samples of the luma plane                    */
    int64_t u_offset, v_offset;  /* size and size * 3 / 2 (samples)                                 */
    int32_t pitch;               /* BGRA row bytes: ((y_stride * 32 + 31) >> 5) << 2                */
    int32_t ctbrows, ctbcols;    /* divceil(height, ctb), divceil(width, ctb)                       */
} ffhip_hevc_layout;
int ffhip_hevc_picture_layout(int pic_width, int pic_height, int ctb_log2, ffhip_hevc_layout *out);

typedef struct ffhip_heif_grid {
    uint8_t version, flags;
    uint16_t rows, cols;
    uint32_t output_width, output_height;
} ffhip_heif_grid;
int ffhip_heif_grid_parse(const uint8_t *item, size_t length, ffhip_heif_grid *out);
int ffhip_heif_grid_compose(uint8_t *d_canvas, int64_t canvas_pitch, int out_w, int out_h,
                            const uint8_t *d_tiles, int64_t tile_pitch, int64_t tile_stride,
                            int tile_w, int tile_h, int rows, int cols, void *stream);

/* ---- VP8 (WebP lossy) residual stage, batched over macroblocks ----
 * Replaces, for n_mb macroblocks at once, what vp8_decode_residual_block does between
 * the token parse and the predictor (format/webp.c:1147-1196): dequantisation (the
 * `absValue * quant` int16 store of webp.c:1061), IWHT_long / IWHT_fast of the Y2 block
 * (webp.c:1067-1106) and idct_4x4_16 (utils/idct.c:100-151) of every block that has
 * more than one token or a non-zero DC (webp.c:1172,1188).  DEVICE pointers:
 *   d_levels   int16 [n_mb][25][16]  quantised levels at their raster position (zig-zag
 *              placement done); blocks 0-15 Y, 16-19 U, 20-23 V, 24 Y2
 *   d_mbinfo   uint8 [n_mb][32]      [0..24] token count per block (the return value of
 *              vp8_get_coefficients), [25] 1 if intra_y_mode != B_PRED (has Y2),
 *              [26] segment id (0..3)
 *   d_quant    uint16 [4][8]         per segment y1_dc,y1_ac,y2_dc,y2_ac,uv_dc,uv_ac,0,0
 *              (struct WEBP_decoder, format/webp.h:276-287)
 *   d_residual int16 [n_mb][384]     the `coeffs` array vp8_prerdict_mb consumes
 * d_levels and d_residual 16-byte aligned, d_mbinfo and d_quant 4-byte aligned. */
int ffhip_vp8_residual_batch(long long n_mb, const int16_t *d_levels, const uint8_t *d_mbinfo,
                             const uint16_t *d_quant, int16_t *d_residual, void *stream);

/* ---- VP8 in-loop deblocking filter for batches of key frames ----
 * The second MB loop of vp8_decode (format/webp.c:1856-1866): loopfilter() (webp.c:1686-1752)
 * with its simple and normal filters (webp.c:1480-1684) on the 8-bit Y/U/V planes that
 * ffhip_vp8_predict_recon wrote, before ffhip_yuv420_to_bgra.
 *   filter_type  0 none, 1 simple, 2 normal  (webp.c:1852-1853)
 *   d_modes      the same [n_images][n_mb][20] records; [0] intra_y_mode, [18] segment_id
 *   d_filters    uint8 [4 segments][2 (i16x16, i4x4)][3] = sub_limit, inter_limit, hev_thresh
 *                (struct vp8_filter as calculate_filter_control_parameter leaves it,
 *                webp.c:1756-1803, format/webp.h:289-293) */
/* The frame-header fields calculate_filter_control_parameter (format/webp.c:1756-1803) reads, as the reference's
 * header parser leaves them (format/webp.h:160-230), and the derivation itself on the host: the per-segment
 * {sub_limit, inter_limit, hev_thresh} triples ffhip_vp8_loopfilter takes as d_filters, and the filter_type
 * argument (0 none, 1 simple, 2 normal; webp.c:1757-1759, 1852-1853). */
typedef struct ffhip_vp8_filter_header {
    uint8_t filter_type;          /* frame header bit: 1 = simple filter, 0 = normal                    */
    uint8_t loop_filter_level;    /* 0..63, 0 = no filtering                                            */
    uint8_t sharpness_level;      /* 0..7                                                               */
    uint8_t segmentation_enabled; /* segmentation.segmentation_enabled                                  */
    uint8_t segment_feature_mode; /* 1: lf_update_value is absolute, 0: a delta to loop_filter_level    */
    int8_t lf_update_value[4];    /* segmentation.lf[s].lf_update_value                                 */
    uint8_t loop_filter_adj_enable;
    int8_t mode_ref_lf_delta0;    /* mb_lf_adjustments.mode_ref_lf_delta_update[0] (intra frame)        */
    int8_t mb_mode_delta0;        /* mb_lf_adjustments.mb_mode_delta_update[0] (B_PRED macroblocks)     */
    uint8_t nbr_partitions;       /* 1, 2, 4 or 8: the reference derives the triples inside its loop over the DCT
                                     PARTITIONS (webp.c:1905-1915), so only segments 0 .. nbr_partitions-1 get any; the
                                     others keep zeros = "no filtering" (a reference defect, kept; its write past
                                     filters[3] with 8 partitions is not reproduced)                            */
} ffhip_vp8_filter_header;
/* host only; filters = uint8 [4 segments][2 (i16x16, i4x4)][3], zeroed first like the reference's calloc'ed decoder */
int ffhip_vp8_filter_params(const ffhip_vp8_filter_header *hdr, uint8_t *filters /* [4][2][3] */, int *filter_type);
int ffhip_vp8_loopfilter(int mbcols, int mbrows, int n_images, int filter_type, const uint8_t *d_modes,
                         const uint8_t *d_filters, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                         int64_t plane_stride_y, int64_t plane_stride_uv, void *stream);
/* ffhip_vp8_predict_recon followed by ffhip_vp8_loopfilter as ONE call (the frame loop of format/webp.c:1833-1866: predict
 * every macroblock, then filter the frame): same arguments, same bytes.  The two row kernels run side by side -- the filter
 * on a stream of the library's own, forked from and joined back into `stream` -- with the filter's rows following the
 * prediction's through its per-row progress counters (a macroblock is filtered once the prediction has finished its right
 * neighbour in the row below: the prediction reads reconstructed, not filtered, samples), so the two dependency chains
 * overlap instead of adding up.  filter_type 0 = prediction only.  FFHIP_VP8_FUSE=0: one after the other.
 * The call is REPEATED by ffhip_stream_sync when it could not finish (batches of up to 2^17 macroblocks): on a device shared with other
 * work one of the two kernels can be kept from becoming resident next to the other, and a bounded wait then runs out.  The two kernels report
 * that in a pinned word of the CALL's own (nobody else's abort sets the retry off); the library keeps a copy of the one thing of the planes'
 * former contents the prediction reads (their last luma column, for the wrapped H_PRED read of predict.c:346-353) and of the call's arguments;
 * ffhip_stream_sync on `stream` then restores that column, runs prediction and filter one after the other, waits, and returns FFHIP_RETRIED
 * (> 0) with the bytes of an undisturbed call in the planes -- or FFHIP_EIO when the repeat failed too, or when the call is no longer the last
 * VP8 prediction / filter call enqueued on `stream`.  FFHIP_RETRIED is not FFHIP_OK on purpose: whatever the CALLER enqueued on `stream` behind
 * the call (a copy, a kernel of its own, ffhip_yuv420_to_bgra) has already consumed the planes of the aborted run and must be enqueued again;
 * the library knows this only of its own stages (ffhip_vp8_decode_frames repeats its colour conversion as part of the retry, and still says
 * FFHIP_RETRIED for the sake of what the caller put behind IT).  Contract: the call's inputs (d_modes, d_residual, d_resmap, d_filters) and
 * planes stay valid and unchanged until ffhip_stream_sync(stream) has returned.  FFHIP_VP8_NO_RETRY=1: FFHIP_EIO, planes unspecified, as
 * before round 4. */
int ffhip_vp8_predict_loopfilter(int mbcols, int mbrows, int n_images, const uint8_t *h_modes, const uint8_t *d_modes,
                                 const int16_t *d_residual, int64_t residual_stride, const int32_t *d_resmap,
                                 int filter_type, const uint8_t *d_filters, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                                 int64_t plane_stride_y, int64_t plane_stride_uv, void *stream);

/* ---- HEVC residual stage, batched over transform units of one size ----
 * For n_tu TUs of size nTbS x nTbS (4, 8, 16 or 32): scale_transform_coefficients
 * (coding/hevc.c:3743-3816) followed by transform_scaled_coeffients (hevc.c:3888-3956,
 * 1-D kernels of hevc.c:3819-3885), i.e. the non-bypass branch of scale_and_transform
 * (hevc.c:4224-4240); the bypass and transform-skip branches (hevc.c:4209-4236) are
 * selected per TU.  DEVICE pointers:
 *   d_level    int16 [n_tu][nTbS*nTbS]  TransCoeffLevel, row-major x + y*nTbS (the layout
 *              of the reference's d[] / r[]; its TransCoeffLevel[cIdx][x][y] is x-major)
 *   d_tuinfo   uint8 [n_tu][4]   [0] qP, [1] flags: 1 = luma intra 4x4 -> idct_4x4_hevc
 *              (DST-VII, utils/idct.c:36-55), 2 = transform_skip_flag, 4 = cu_transquant_bypass,
 *              8 = rotateCoeffs; [2] scaling matrixId (0..5); [3] 0
 *   d_scaling  uint8 [6][nTbS*nTbS] ScalingFactor[sizeId][matrixId] row-major, or NULL for
 *              scaling_list_enabled_flag == 0 (m = 16)
 *   bitdepth   BitDepthY or BitDepthC of the component the TUs belong to; epp =
 *              extended_precision_processing_flag
 *   d_residual int16 [n_tu][nTbS*nTbS]  r[] as construct_pic_pior_to_filtering consumes it */
int ffhip_hevc_residual_batch(int nTbS, long long n_tu, const int16_t *d_level, const uint8_t *d_tuinfo,
                              const uint8_t *d_scaling, int bitdepth, int epp, int16_t *d_residual,
                              void *stream);

/* ---- VP8 intra prediction + residual add for batches of key frames ----
 * What vp8_prerdict_mb does for every macroblock of vp8_decode's loop (format/webp.c:1833-1851):
 * pred_luma + pred_chrome (format/predict.c:426-645) with the residual of
 * ffhip_vp8_residual_batch, written into 8-bit Y/U/V planes of stride 16*mbcols / 8*mbcols.
 *   h_modes / d_modes  the SAME uint8 [n_images][mbrows*mbcols][20] records on the host (used
 *                      to schedule the dependency wavefronts) and on the device:
 *                      [0] intra_y_mode (0 DC, 1 TM, 2 V, 3 H, 4 B_PRED), [1] intra_uv_mode,
 *                      [2..17] imodes[16] (4x4 modes 0..9), [18..19] 0  (format/webp.h:243-256).
 *                      A y / uv mode out of range -- or, in a B_PRED record, a 4x4 mode above 9 (the
 *                      reference indexes a table of ten predictors with it) -- is FFHIP_EINVAL: from this call for batches of up
 *                      to 2^17 macroblocks (checked on the host copy), for larger ones from the next
 *                      ffhip_stream_sync (checked by a kernel in front of the prediction: the planes
 *                      are then left untouched)
 *   d_residual         int16 [n_images][.][384], image i at + i*residual_stride (elements)
 *   d_resmap           int32 [n_images][n_mb] residual row used by each macroblock, or NULL for
 *                      the identity; the reference keeps the PREVIOUS macroblock's coefficients
 *                      when mb_skip_coeff is set (webp.c:1207-1223) -- map skipped MBs there
 *   d_y/d_u/d_v        planes, image i at + i*plane_stride_*; their initial contents and the
 *                      bytes before them matter exactly where the reference's 16x16 V_PRED /
 *                      H_PRED read raw memory at the top row / left column (predict.c:338-353):
 *                      bytes before a plane read as 0.
 * Enqueues ONE launch on `stream`: a wave per macroblock row, rows chained through progress
 * counters inside the launch, as many waves as the device can hold at once -- the
 * frames of a batch run side by side, so throughput grows with the batch up to a few hundred frames; should a wave's bounded wait ever run out, the next
 * ffhip_stream_sync on any stream returns FFHIP_EIO.  FFHIP_VP8_PRED_MODE=levels selects the older
 * one-launch-per-wavefront-level form.  Scratch is kept per stream: calls on different streams (or
 * host threads with their own streams) may be in flight together; calls on one stream are ordered. */
int ffhip_vp8_predict_recon(int mbcols, int mbrows, int n_images, const uint8_t *h_modes,
                            const uint8_t *d_modes, const int16_t *d_residual, int64_t residual_stride,
                            const int32_t *d_resmap, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                            int64_t plane_stride_y, int64_t plane_stride_uv, void *stream);

/* ---- the VP8 key-frame chain of a batch as ONE call: prediction + loop filter + colour conversion ----
 * The frame loop of vp8_decode (format/webp.c:1833-1868: vp8_prerdict_mb for every macroblock, loopfilter for every
 * macroblock, YUV420_to_BGRA32) with the residual of ffhip_vp8_residual_batch; the same bytes as
 * ffhip_vp8_predict_loopfilter on zero-initialised planes followed by ffhip_yuv420_to_bgra.  Arguments as there:
 *   h_modes / d_modes, d_residual, residual_stride, d_resmap   as ffhip_vp8_predict_recon (h_modes may be NULL for
 *                      batches of more than 2^17 macroblocks: those are checked on the device)
 *   filter_type, d_filters                                      as ffhip_vp8_loopfilter
 *   d_bgra, pitch, image_stride                                 as ffhip_yuv420_to_bgra: 16*mbrows rows of 16*mbcols pixels
 *   d_y / d_u / d_v    NULL, or planes (stride 16*mbcols / 8*mbcols, image i at + i*plane_stride_*) that receive the
 *                      filtered samples as well.  The planes are OUTPUTS only here: where the reference's 16x16 H_PRED reads
 *                      raw memory left of a row's first pixel (predict.c:346-353) it finds the last pixel of the row above
 *                      and, below it, samples not reconstructed yet -- 0, as in the freshly allocated planes of vp8_decode.
 * Batches of at least half as many frames as the device has CUs run as ONE kernel in which a workgroup owns a frame, its
 * waves the frame's macroblock rows, and a wave predicts, filters and converts its macroblock before anything is stored
 * (every pixel is written once, as BGRA); smaller batches run the three stages (row kernels, then the colour
 * kernel: a single frame's critical path is shorter there).  FFHIP_VP8_FRAMES=fused|rows forces either. */
/* which form ffhip_vp8_decode_frames takes for a batch of n_images on the current device: 1 the frame kernel, 0 the row kernels + colour kernel
 * (half the device's compute units and more take the frame kernel; FFHIP_VP8_FRAMES / FFHIP_VP8_FRAMES_MIN move that) */
int ffhip_vp8_decode_frames_form(int n_images);
int ffhip_vp8_decode_frames(int mbcols, int mbrows, int n_images, const uint8_t *h_modes, const uint8_t *d_modes,
                            const int16_t *d_residual, int64_t residual_stride, const int32_t *d_resmap, int filter_type,
                            const uint8_t *d_filters, uint8_t *d_bgra, int pitch, int64_t image_stride, uint8_t *d_y,
                            uint8_t *d_u, uint8_t *d_v, int64_t plane_stride_y, int64_t plane_stride_uv, void *stream);

/* ---- HEVC intra prediction + reconstruction for a list of transform units ----
 * decode_intra_block steps 5-10 (coding/hevc.c:4730-4790) for every TU of a picture:
 * intra_sample_prediction (hevc.c:4542-4662: neighbour gathering, reference_sample_substitution
 * :4277-4351, filtering_neighbouring_samples :4355-4426, hevc_intra_planar/DC/angular
 * format/predict.c:651-792), the optional rdpcm residual modification (hevc.c:3960-3977) and
 * construct_pic_pior_to_filtering (hevc.c:4252-4274) into int16 sample planes laid out like
 * the reference's picture (hevc.c:7225-7230).  TUs are listed in decode order; what the
 * reference derives while parsing (neighbour availability by z-scan order / slice / tile,
 * hevc.c:4570-4608) arrives here as bit masks.
 * Order contract: inside each PLANE a TU follows every TU its availability bits point at; how the planes
 * interleave is free (they never read each other here).  The reference's own order -- per coding unit the luma
 * tree, then Cb, then Cr (decode_cu_coded_intra_prediction_mode, hevc.c:5013-5180) -- is taken as it comes:
 * scheduling runs are defined on each plane's own subsequence, and a list that switches planes inside a
 * scheduling window is sorted by plane (a stable device-side copy) in front of the planner. */
typedef struct ffhip_hevc_tu {
    uint16_t x, y;       /* top-left of the TU in samples of its component plane           */
    uint8_t log2_size;   /* 2..5                                                            */
    uint8_t cidx;        /* 0 Y, 1 Cb, 2 Cr                                                 */
    uint8_t pred_mode;   /* predModeIntra: 0 planar, 1 DC, 2..34 angular                    */
    uint8_t flags;       /* FFHIP_TU_*                                                      */
    uint32_t res_offset; /* element offset of the TU's residual block in d_residual         */
    int32_t res_scale;   /* ResScaleVal of 8.6.6 (hevc.c:3494-3497: 0, +-1, +-2, +-4, +-8); used with FFHIP_TU_CCP */
    uint64_t avail_top;  /* bit k: neighbour (x+k, y-1), k = 0..2n-1, is available          */
    uint64_t avail_left; /* bit k: neighbour (x-1, y+k) is available                        */
} ffhip_hevc_tu;
#define FFHIP_TU_CORNER 0x01   /* neighbour (x-1, y-1) available                             */
#define FFHIP_TU_RESIDUAL 0x02 /* numSigCoeff != 0: add the residual block (hevc.c:4737)      */
#define FFHIP_TU_FILTER 0x04   /* 8.4.4.2.3 applies: intra_smoothing_disabled_flag == 0 and
                                  (cIdx == 0 or ChromaArrayType == 3)  (hevc.c:4629-4633)    */
#define FFHIP_TU_STRONG 0x08   /* sps strong_intra_smoothing_enabled_flag                    */
#define FFHIP_TU_NO_BF 0x10    /* disableIntraBoundaryFilter (hevc.c:4642-4648)              */
#define FFHIP_TU_NO_DC_BF 0x20 /* intra_boundary_filtering_disabled_flag (DC edge filter)    */
#define FFHIP_TU_RDPCM 0x40    /* residualDpcm == 1: 8.6.5 on the residual before the add    */
#define FFHIP_TU_CCP 0x80      /* 8.6.6 cross-component prediction after 8.6.5, exactly as the reference
                                  calls it (hevc.c:4750-4756): the "luma" residual it passes is the chroma
                                  block itself, so r += (res_scale * ((r << BitDepthC) >> BitDepthY)) >> 3 */
/* h_tus / d_tus: the SAME n_tus records on the host (dependency scheduling) and on the device.
 * d_residual: int16 residual blocks (row-major n*n each) as ffhip_hevc_residual_batch writes
 * them.  Planes: int16, strides in samples; d_cb/d_cr may be NULL for 4:0:0.  ONLY ENQUEUES: h_tus is validated on
 * the host (and the scheduling window chosen from it) -- record by record up to 2^17 TUs; of a larger list the host
 * looks at a sample only (every 64th stretch of 4096 records, every 256th from a million records on) and EVERY record of d_tus is checked by a kernel in front
 * of everything else: a bad record found there refuses the call through the stream (the call returns 0, nothing is
 * written, the next ffhip_stream_sync returns FFHIP_EINVAL; FFHIP_HEVC_HOST_CHECK=1 keeps the whole check on the host) --,
 * the schedule is built on the device (hand-written kernels: no library primitive) and ONE launch follows --
 * TUs grouped by 32x32 window, a wave per group, done flags between groups -- which reads the
 * planner's verdict itself: a list it refuses is decoded by one wave in decode order inside the same launch (slow,
 * exact).  A bounded wait that ever runs out surfaces as FFHIP_EIO from the next ffhip_stream_sync.
 * FFHIP_HEVC_INTRA_MODE=levels selects the older one-launch-per-dependency-level form and FFHIP_HEVC_PLAN=host the
 * host-side planner (both synchronise the stream).  Scratch is kept per stream, as for VP8. */
/* Host only, no device needed: the group schedule ffhip_hevc_intra_recon builds for an already valid
 * list -- out_ticket[i] = ticket of the group of TU i, out_wait[i] = TUs of other groups it waits for
 * (either may be NULL), stats[4] = {groups, luma window log2 used, wait entries, TUs served from the
 * LDS tile}; window_log2 0 = the default.  FFHIP_EINVAL when no window gives a deadlock-free order. */
int ffhip_hevc_intra_plan(const ffhip_hevc_tu *h_tus, long long n_tus, int width_y, int height_y,
                          int width_c, int height_c, int window_log2, uint32_t *out_ticket,
                          uint32_t *out_wait, int32_t *stats);
/* Diagnostics: what the DEVICE planner made of the list of this thread's last ffhip_hevc_intra_recon call (which it waits for):
 * out[0] != 0 the plan was refused and the list decoded by the one-wave serial kernel (exact, slow); out[1] groups; out[3] != 0 the
 * tickets are in decode order (no coding-tree wavefront found); out[4] the widest wavefront; out[5] log2 of the luma scheduling window;
 * out[6] != 0 a record failed the device's validation; out[7] != 0 the list was sorted by plane first (it interleaves the planes inside a
 * window, as the reference's order does).  FFHIP_EINVAL when that call did not use the device planner. */
int ffhip_debug_hevc_plan_result(uint32_t out[8]);
int ffhip_hevc_intra_recon(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus,
                           const int16_t *d_residual, int16_t *d_y, int16_t *d_cb, int16_t *d_cr,
                           int width_y, int height_y, int y_stride, int width_c, int height_c,
                           int uv_stride, int bitdepth_y, int bitdepth_c, void *stream);

/* The same for a list that is the CONCATENATION of independent pictures sharing one plane set -- the tiles of a HEIF grid, decoded one after
 * the other by the tile loop of format/heif.c:297-309 -- tile k being records [tile_first[k], tile_first[k + 1]) (tile_first[0] = 0; the last tile
 * ends at n_tus).  Same samples as ffhip_hevc_intra_recon on the whole list.  What the call adds is a PIPELINE across the stages of a decoder:
 * the pre-pass (validation, planner, substitution table, per-pixel programs: a third of an eight-picture call) depends on the TU list alone, so it
 * runs on a stream of the library's own WITHOUT waiting for `stream` -- next to whatever the caller enqueued there in front of this call: the
 * residual batches of these tiles, the colour conversion of the picture before -- and only the grouped kernel takes its place in `stream`.
 * Contract: d_tus holds the records WHEN THE CALL IS MADE (uploaded by a blocking copy, or by a copy whose event the host has waited for), not merely
 * by work enqueued on `stream`; h_tus / d_tus stay untouched until `stream` has passed the call.  d_residual and the planes are read and written in
 * `stream` order as always.  Consecutive calls alternate between two sets of library scratch, so the pre-pass of call n + 1 also runs next to the
 * tail of call n's grouped kernel (FFHIP_HEVC_TILE_SCRATCHES=1: one set).  FFHIP_HEVC_TILE_EARLY=0: everything in `stream` order
 * (= ffhip_hevc_intra_recon).  Eight / four / one 8K picture(s) as grids of 135 tiles, whole chain: 103 / 95 / 57 Gpixel/s against 90 / 85 / 48.
 * FFHIP_HEVC_TILE_CHUNKS=2..4 cuts the list at tile boundaries into chunks whose pre-passes and grouped kernels overlap on streams of the library's
 * own (the caller's word that no tile references another makes that legal): built, bit-exact, and measured SLOWER than the one launch (DESIGN.md
 * 4.7 "Round 5") -- a tested switch, not the default. */
int ffhip_hevc_intra_recon_tiles(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus,
                                 const long long *tile_first, int n_tiles, const int16_t *d_residual, int16_t *d_y,
                                 int16_t *d_cb, int16_t *d_cr, int width_y, int height_y, int y_stride, int width_c,
                                 int height_c, int uv_stride, int bitdepth_y, int bitdepth_c, void *stream);

/* ffhip_hevc_intra_recon_tiles and the colour conversion of the plane set (YUV420_to_BGRA32_16bit, utils/colorspace.c:628-669, as hevc.c:7260-7270
 * calls it) as ONE call: the tile loop of format/heif.c:297-309 with the conversion behind it.  d_bgra: pixel (x, y) of the plane set at
 * d_bgra + y * pitch + 4 * x (width_y x height_y pixels; 4:2:0: width_c = width_y / 2, height_c = height_y / 2; width_y a multiple of 4, height_y
 * even).  The colour kernel follows the grouped kernel on `stream`, next to the NEXT call's pre-pass on the library's stream.  One tile
 * (n_tiles = 1, tile_first = {0}) is an ordinary picture. */
int ffhip_hevc_decode_tiles(const ffhip_hevc_tu *h_tus, const ffhip_hevc_tu *d_tus, long long n_tus, const long long *tile_first,
                            int n_tiles, const int16_t *d_residual, int16_t *d_y, int16_t *d_cb, int16_t *d_cr, int width_y,
                            int height_y, int y_stride, int width_c, int height_c, int uv_stride, int bitdepth_y, int bitdepth_c,
                            uint8_t *d_bgra, int64_t pitch, void *stream);

/* ---- host-side JPEG front end and BMP sink (plain C, no GPU) ----
 * ffhip_jpeg_probe / ffhip_jpeg_entropy_decode stand where the marker loop, read_dqt,
 * read_compressed_scan and decode_data_unit stand (format/jpg.c:78-105, 255-415, 588-655,
 * 771-855; coding/huffman.c:92-222): a baseline / extended-sequential Huffman scan becomes the
 * MCU-order coefficient planes and natural-order quant tables ffhip_jpeg_recon_batch reads.
 * Progressive, arithmetic, 12-bit, non-interleaved or chroma-subsampling-other-than-1x1 files
 * return FFHIP_EINVAL (keep the C path).  All pointers are HOST pointers. */
int ffhip_jpeg_probe(const uint8_t *file, size_t len, ffhip_jpeg_geom *geom, int *width, int *height);
int ffhip_jpeg_entropy_decode(const uint8_t *file, size_t len, const ffhip_jpeg_geom *expect,
                              int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint16_t *quant /* [4][64] */);
/* the same for one picture with n_threads host threads: a file with a DRI segment has independent
 * restart intervals (jpg.c:562-573), which are shared out; without one it is a single-thread decode */
int ffhip_jpeg_entropy_decode_mt(const uint8_t *file, size_t len, const ffhip_jpeg_geom *expect,
                                 int16_t *coef_y, int16_t *coef_u, int16_t *coef_v, uint16_t *quant,
                                 int n_threads);
/* n files of one geometry over n_threads host threads (with at least twice as many threads as files
 * the threads work inside each picture instead, as above); image i writes planes at
 * + i*blocks*64 and quant at + i*256; status[i] receives each file's code. */
int ffhip_jpeg_entropy_batch(const uint8_t *const *files, const size_t *lens, int n, int n_threads,
                             const ffhip_jpeg_geom *geom, int16_t *coef_y, int16_t *coef_u,
                             int16_t *coef_v, uint16_t *quant, int *status);
/* Test hook for the host staging pass of ffhip_jpeg_entropy_batch_gpu (needs no device): unstuffs the entropy-coded
 * bytes src[0..len) into dst (FF 00 -> FF), pads every restart interval to 4 bytes + 4 zero bytes, writes the clean
 * offset of interval k to seg[k] (k < n_seg) and the clean length to *clean_len; returns the number of intervals
 * found (RSTn-separated; another marker or the n_seg-th RSTn ends the scan), or FFHIP_EINVAL.
 * dst must hold len + 8 * n_seg + 64 bytes.  The reference's counterpart is the byte loop of read_compressed_scan
 * (format/jpg.c:588-637). */
int ffhip_jpeg_stage_scan_test(uint8_t *dst, const uint8_t *src, size_t len, uint32_t *seg, uint32_t n_seg, size_t *clean_len);
/* the same, and raw[k] = the bytes of interval k without its padding (what the subsequence decoder cuts into lanes of 2048 / 4096 / 8192 bits) */
int ffhip_jpeg_stage_scan_raw_test(uint8_t *dst, const uint8_t *src, size_t len, uint32_t *seg, uint32_t n_seg, size_t *clean_len, uint32_t *raw);
/* Test hook (needs no device): the two-level look-up table the device Huffman kernels use for table `which` (0..3 DC, 4..7 AC) of a file, 1536 uint16:
 * [0..511] by the next 9 bits: (length << 8) | symbol, or 0x8000 | g for a prefix of longer codes; [512 + 128 g + b] group g by the 7 bits behind
 * the prefix; 0x5000 = no code starts with these bits (a table that holds all its codes says so itself); 0 = take the canonical-code walk.  The
 * reference's counterpart is huffman_decode_symbol (coding/huffman.c:92-222). */
int ffhip_jpeg_lut_test(const uint8_t *file, size_t len, int which, uint16_t *out);

/* The same front end ON the device: decodes straight into DEVICE planes (d_coef_*, d_quant [n][4][64]) laid out for
 * ffhip_jpeg_recon_batch with quant_stride 256; the host only parses headers, finds the RSTn markers and unstuffs the
 * bytes into pinned memory.  files/lens/status are HOST arrays.  Round 5: a lane decodes one SUBSEQUENCE of a restart interval
 * -- of the whole scan, in a file without DRI; 2048, 4096 or 8192 bits, by the bits an MCU takes in the batch (up to 1024, up to 2048, more;
 * FFHIP_JPEG_SYNC_BITS sets it), worked out once per call -- and the lanes are brought into step with each other over a few rounds
 * (Huffman-coded data self-synchronises; DESIGN.md 5 "The subsequence decoder"), so files need no restart markers to decode
 * in parallel, and a batch may mix files with and without.  FFHIP_JPEG_SYNC=0: the kernel of rounds 3-4, one lane per restart
 * interval (a file without DRI is ONE lane's then: for batches of a thousand files or more only); unset, that kernel also takes
 * the batches whose restart intervals are a subsequence or two long by that same figure (a DRI of a few MCUs); =1 keeps the subsequence decoder on those.
 * FFHIP_EINVAL for a file of another geometry, and for a damaged or truncated scan (status[] says which picture; nothing of
 * the batch is to be used then -- ffhip_jpeg_decode_files* fall back to the host decoder).  Synchronises `stream` (the
 * per-picture verdicts come back with it). */
int ffhip_jpeg_entropy_batch_gpu(const uint8_t *const *files, const size_t *lens, int n, int n_threads /* host: header
                                 parsing, marker search, staging */, const ffhip_jpeg_geom *geom, int16_t *d_coef_y, int16_t *d_coef_u, int16_t *d_coef_v, uint16_t *d_quant,
                                 int *status, void *stream);

/* Diagnostics: where the calling thread's last ffhip_jpeg_entropy_batch_gpu call spent its time, in microseconds: out[0] header parse,
 * [1] layout, [2] unstuffing + marker search into pinned memory (each part's upload and kernels are enqueued behind it), [3] tables, [4] enqueue,
 * [5] the wait for uploads + clears + kernels, [6] HIP events on the call's stream around everything the call has the device do (the subsequence
 * decoder: uploads waited for, rounds, scan, write pass of all parts; FFHIP_JPEG_SYNC=0: the Huffman kernel alone), [7] the whole call. */
int ffhip_debug_huff_times(double out[8]);

/* Files in, pixels out (f1 + the hot path + f2's producer side): n baseline JPEG files of ONE geometry are
 * entropy-decoded `chunk` pictures at a time (0 = default: 32 pictures, or as many small ones as make 256 MB of BGRA; 8 with the
 * host decoder) -- on the device (ffhip_jpeg_entropy_batch_gpu), or, when that
 * refuses a chunk or FFHIP_JPEG_GPU_ENTROPY=0 says so, by n_threads host threads into pinned memory -- while the
 * previous chunk is copied to the device, reconstructed by one launch and copied back -- a double-buffered
 * pipeline whose steady state is the slower of host entropy decode and PCIe.  bgra is HOST memory,
 * pixel (x, y) of picture i at bgra + i*image_stride + y*pitch + 4*x (coded size, geom_out tells it);
 * status[i] = per-file code, the return value the first failure.  The layout format/jpg.c:851-852 hands
 * to struct pic.  Not for single-component files with several blocks per MCU.  A destination in pinned
 * memory (ffhip_host_malloc, or the caller's own hipHostMalloc / hipHostRegister) receives the device copy
 * directly; a pageable one goes through pinned staging and a threaded copy.  Buffers are kept between calls;
 * one call at a time. */
/* The same with the pixels left ON THE DEVICE (d_bgra, pitch and image_stride as for ffhip_jpeg_recon_batch): for a
 * consumer that lives on the GPU only the compressed bytes cross PCIe.  Files with or without restart markers: one batch, entropy
 * decode on the device (ffhip_jpeg_entropy_batch_gpu: the subsequence decoder by default), each part's reconstruction enqueued behind its
 * write pass (the entropy stage synchronises the stream).  Only when the device decoder refuses the batch with FFHIP_EINVAL (or
 * FFHIP_JPEG_GPU_ENTROPY=0 says so) do host threads decode chunks of a few pictures into pinned memory while the previous chunk is
 * uploaded and reconstructed on a stream of the library's own; `stream` is synchronised first and every picture is in d_bgra when the
 * call returns. */
int ffhip_jpeg_decode_files_device(const uint8_t *const *files, const size_t *lens, int n, int n_threads,
                                   ffhip_jpeg_geom *geom_out, uint8_t *d_bgra, int64_t pitch, int64_t image_stride,
                                   int *status, void *stream);
void *ffhip_host_malloc(size_t bytes); /* pinned host memory */
void ffhip_host_free(void *p);
int ffhip_jpeg_decode_files(const uint8_t *const *files, const size_t *lens, int n, int n_threads, int chunk,
                            ffhip_jpeg_geom *geom_out, uint8_t *bgra, int64_t pitch, int64_t image_stride,
                            int *status);

/* display/bmpwriter.c:19-81: 54-byte header + top-down 32-bit rows; byte-identical files. */
int ffhip_bmp_write(const char *path, const uint8_t *bgra, int width, int height, int64_t pitch);

/* Device-to-device copy kernel (16 B/lane, grid-stride) used by bench.py to
 * calibrate the achievable HBM rate next to the fused kernel. */
int ffhip_copy_calibrate(void *d_dst, const void *d_src, size_t bytes, void *stream);
/* A second calibration for the same purpose: the 4:2:0 fused kernel's own ACCESS PATTERN -- its loads and its stores, same grid, same addresses --
 * with the arithmetic taken out, on the caller's buffers (d_bgra receives meaningless bytes).  The rate of this launch is the ceiling the fused
 * kernel can be read against on exactly this placement of its buffers (DESIGN.md 5 "Round 6").  Arguments as ffhip_jpeg_recon_batch; every layout a fused kernel takes
 * (4:2:0, 4:4:4, 4:2:2, 4:4:0, 4:1:1, its transpose, grey), FFHIP_EINVAL for the others. */
int ffhip_jpeg_pattern_calibrate(const ffhip_jpeg_geom *geom, int n_images, const int16_t *d_coef_y, const int16_t *d_coef_u, const int16_t *d_coef_v,
                                 const uint16_t *d_quant, int64_t quant_stride, uint8_t *d_bgra, int64_t pitch, int64_t image_stride, void *stream);

/* ---- batches over the GPUs of one node, from C (ffhip_shard.hip) ----
 * The reference decodes one image at a time on one thread (format/jpg.c:458-585) and has no collective of any kind
 * (SURVEY 2.1); images are independent, so a batch shards into contiguous image ranges -- one process and one GPU
 * per rank -- with no data-path exchange.  The only collective is the batch close: ONE ncclAllGather (RCCL over
 * xGMI) of a 32-byte record per rank, which is also the batch barrier.  RCCL is bound at run time (dlopen; a copy
 * already in the process is reused), so one-GPU callers never need it. */
typedef struct ffhip_batch_record {
    int32_t rank;      /* who                                                                   */
    int32_t status;    /* 0, or the FFHIP_E* code of the rank's first failing call              */
    int64_t first;     /* the rank's image range [first, first + count)                         */
    int64_t count;
    uint64_t checksum; /* sum of the rank's per-image checksums (ffhip_bgra_checksum), mod 2^64 */
} ffhip_batch_record;
#define FFHIP_COMM_ID_BYTES 128
/* contiguous image range of `rank` out of `world`; sizes differ by at most one */
int ffhip_shard_range(long long n_images, int rank, int world, long long *first, long long *count);
/* rank 0: a fresh communicator id (ncclGetUniqueId); the host program hands its 128 bytes to the other ranks */
int ffhip_comm_unique_id(void *id128);
/* every rank, on its own device (after ffhip_init): ncclCommInitRank; NULL on failure */
void *ffhip_comm_init_rank(const void *id128, int rank, int world);
void ffhip_comm_destroy(void *comm);
/* Enqueues the all-gather of this rank's record behind what `stream` holds, synchronises `stream` and fills
 * h_records[world] in rank order.  comm == NULL with world == 1: the one-GPU case (stream sync + own record). */
int ffhip_batch_close(void *comm, int rank, int world, long long first, long long count, int status,
                      uint64_t checksum, ffhip_batch_record *h_records, void *stream);
/* host only: 1 when the records tile [0, n_images) exactly, record r is rank r's and every status is 0 */
int ffhip_batch_complete(const ffhip_batch_record *records, int world, long long n_images);
/* per image the sum over its 32-bit pixels p[i] (row-major over width x height) of p[i] * ((i & 0xffff) + 1), mod
 * 2^64, into d_sums[n_images] (device): lets a rank vouch for gigabytes of output with 8 bytes per image */
int ffhip_bgra_checksum(const uint8_t *d_bgra, int64_t pitch, int64_t image_stride, int width, int height,
                        int n_images, uint64_t *d_sums, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FFPIC_HIP_H */
