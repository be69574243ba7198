/*
 * ffhip_color.hip -- planar YUV -> BGRA converters of the WebP and HEVC back ends.
 *   ffhip_yuv420_to_bgra      == YUV420_to_BGRA32        utils/colorspace.c:291-329 (uint8 planes)
 *   ffhip_yuv420_to_bgra_16   == YUV420_to_BGRA32_16bit  utils/colorspace.c:628-669 (int16 planes)
 *   ffhip_yuv400_to_bgra_16   == YUV400_to_BGRA32_16bit  utils/colorspace.c:715-742
 * The reference walks MB by MB / CTB by CTB; the addressing reduces to a raster with
 * nearest-neighbour chroma (r/2, c/2).  HBM-bound: 1.5 B (uint8) or 3 B (int16) in + 4 B
 * out per pixel.  One lane = 4 pixels x 2 rows (two chroma samples), stores are one
 * dwordx4 per lane per row, 1 KiB contiguous per wave.
 * Compile with -ffp-contract=off.
 */
#include "ffhip_colorterms.h"

struct PlanarArgs {
    const void *y, *u, *v;
    uint8_t *bgra;
    long long y_stride, uv_stride;           /* samples */
    long long plane_y, plane_uv;             /* samples between images */
    long long pitch, image_stride;           /* bytes */
    int width, height, n_images;             /* pixels; width % 4 == 0, height % 2 == 0 */
};

template <typename T>
__device__ __forceinline__ void convert_quad(const T *yp, int u_s, int v_s, bool in_domain_check, u32 out[4])
{
    /* u_s, v_s: raw chroma samples; uu = (int16)(u - 128) as the reference stores it */
    const int uu = (int)(short)(u_s - 128), vv = (int)(short)(v_s - 128);
    const bool dom = !in_domain_check || ((unsigned)u_s <= 8191u && (unsigned)v_s <= 8191u);
    ChromaTerms t = ff_chroma_terms(dom ? uu : 0, dom ? vv : 0);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int yy = (int)yp[k];
        if (dom && !t.sensitive && (!in_domain_check || (unsigned)yy <= 8191u)) out[k] = ff_bgra_int(yy, t);
        else out[k] = ff_bgra_fp64(yy, uu, vv);
    }
}

/* 8-bit planes (WebP): the packed forms of the fused kernels -- the chroma terms of the lane's two chroma samples once for both rows (one fma
 * and one add per term), per pixel pair three 16-bit adds, three saturating packs, three byte permutes: 93 vector instructions per lane
 * where the per-pixel integer form below (kept for the 16-bit planes, whose samples may lie outside the exact domain) took 296.  Same lane
 * layout and stores: 4 pixels x 2 rows per lane, one non-temporal dwordx4 per lane and row. */
__global__ __launch_bounds__(256) void k_yuv420_to_bgra_u8(PlanarArgs a)
{
    const int w4 = a.width / 4, h2 = a.height / 2;
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int img = blockIdx.z;
    if (gx >= w4 || gy >= h2) return;
    const uint8_t *Y = (const uint8_t *)a.y + (long long)img * a.plane_y;
    const uint8_t *U = (const uint8_t *)a.u + (long long)img * a.plane_uv;
    const uint8_t *V = (const uint8_t *)a.v + (long long)img * a.plane_uv;
    const long long co = (long long)gy * a.uv_stride + 2 * gx;
    const uint8_t *y0 = Y + (long long)(2 * gy) * a.y_stride + 4 * gx;
    /* (alignment: 4 * gx and 2 * gx bytes into rows whose stride the entry point checks to be a multiple of 4 / 2) */
    const u32 l0 = *(const u32 *)y0, l1 = *(const u32 *)(y0 + a.y_stride);
    const u32 eu = (u32)*(const unsigned short *)(U + co), ev = (u32)*(const unsigned short *)(V + co);
    const Packed420 p = ff_packed420_terms(eu, ev);
    uint8_t *o = a.bgra + (long long)img * a.image_stride + (long long)(2 * gy) * a.pitch + 16LL * gx;
    __builtin_nontemporal_store(ff_packed420_row(p, l0), (u32x4 *)o);
    __builtin_nontemporal_store(ff_packed420_row(p, l1), (u32x4 *)(o + a.pitch));
}

template <typename T, bool CHECK>
__global__ __launch_bounds__(256) void k_yuv420_to_bgra(PlanarArgs a)
{
    const int w4 = a.width / 4, h2 = a.height / 2;
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);       /* 4-pixel group */
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);        /* row pair */
    const int img = blockIdx.z;
    if (gx >= w4 || gy >= h2) return;
    const T *Y = (const T *)a.y + (long long)img * a.plane_y;
    const T *U = (const T *)a.u + (long long)img * a.plane_uv;
    const T *V = (const T *)a.v + (long long)img * a.plane_uv;
    const long long co = (long long)gy * a.uv_stride + 2 * gx;
    const int u0 = U[co], u1 = U[co + 1], v0 = V[co], v1 = V[co + 1];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const T *yr = Y + (long long)(2 * gy + r) * a.y_stride + 4 * gx;
        T ys[4] = {yr[0], yr[1], yr[2], yr[3]};
        u32 px[4];
        convert_quad<T>(ys, u0, v0, CHECK, px);
        convert_quad<T>(ys + 2, u1, v1, CHECK, px + 2);
        u32x4 o = {px[0], px[1], px[2], px[3]};
        __builtin_nontemporal_store(o, (u32x4 *)(a.bgra + (long long)img * a.image_stride +
                                                 (long long)(2 * gy + r) * a.pitch + 16LL * gx));
    }
}

/* grey: clamp(Y, 255) replicated into all four bytes, alpha included (colorspace.c:731-735) */
__global__ __launch_bounds__(256) void k_yuv400_to_bgra_16(PlanarArgs a)
{
    const int w4 = a.width / 4;
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int img = blockIdx.z;
    if (gx >= w4 || y >= a.height) return;
    const short *yr = (const short *)a.y + (long long)img * a.plane_y + (long long)y * a.y_stride + 4 * gx;
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = (u32)ff_clamp255((int)yr[k]) * 0x01010101u;
    __builtin_nontemporal_store(o, (u32x4 *)(a.bgra + (long long)img * a.image_stride + (long long)y * a.pitch + 16LL * gx));
}

static int planar_args(PlanarArgs &a, const void *y, const void *u, const void *v, int y_stride, int uv_stride,
                       int rows, int cols, int unit, uint8_t *bgra, int pitch, int n_images, int64_t plane_y,
                       int64_t plane_uv, int64_t image_stride, bool chroma)
{
    if (!y || !bgra || (chroma && (!u || !v)) || rows <= 0 || cols <= 0 || n_images < 0) return FFHIP_EINVAL;
    if (unit < 2 || (unit & 1)) return FFHIP_EINVAL;
    a.width = cols * unit; a.height = rows * unit;
    if ((a.width & 3) || y_stride < a.width || (chroma && uv_stride < a.width / 2)) return FFHIP_EINVAL;
    if (pitch < a.width * 4 || (pitch & 15) || ((uintptr_t)bgra & 15) || (image_stride & 15)) return FFHIP_EINVAL;
    if (n_images > 65535) return FFHIP_EINVAL;
    a.y = y; a.u = u; a.v = v; a.bgra = bgra; a.y_stride = y_stride; a.uv_stride = uv_stride;
    a.plane_y = plane_y; a.plane_uv = plane_uv; a.pitch = pitch; a.image_stride = image_stride; a.n_images = n_images;
    return FFHIP_OK;
}

extern "C" int ffhip_yuv420_to_bgra(uint8_t *d_bgra, int pitch, const uint8_t *d_y, const uint8_t *d_u,
                                    const uint8_t *d_v, int y_stride, int uv_stride, int mbrows, int mbcols,
                                    int n_images, int64_t plane_stride_y, int64_t plane_stride_uv,
                                    int64_t image_stride, void *stream)
{
    PlanarArgs a;
    int rc = planar_args(a, d_y, d_u, d_v, y_stride, uv_stride, mbrows, mbcols, 16, d_bgra, pitch, n_images,
                         plane_stride_y, plane_stride_uv, image_stride, true);
    if (rc || n_images == 0) return rc;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    dim3 grid((a.width / 4 + 63) / 64, (a.height / 2 + 3) / 4, n_images);
    /* the packed kernel loads a lane's samples as a dword of luma and a halfword of each chroma plane: strides and bases that keep those aligned
     * (every picture of 16x16 macroblocks does); anything else takes the per-sample form */
    const bool aligned = !((uintptr_t)d_y & 3) && !(y_stride & 3) && !(plane_stride_y & 3) && !((uintptr_t)d_u & 1) && !((uintptr_t)d_v & 1) && !(uv_stride & 1) && !(plane_stride_uv & 1);
    if (aligned && !FFHIP_ENV("FFHIP_COLOR8_SCALAR")) hipLaunchKernelGGL(k_yuv420_to_bgra_u8, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_yuv420_to_bgra<uint8_t, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}

extern "C" int ffhip_yuv420_to_bgra_16(uint8_t *d_bgra, int pitch, const int16_t *d_y, const int16_t *d_u,
                                       const int16_t *d_v, int y_stride, int uv_stride, int ctbrows, int ctbcols,
                                       int ctbsize, int n_images, int64_t plane_stride_y, int64_t plane_stride_uv,
                                       int64_t image_stride, void *stream)
{
    PlanarArgs a;
    int rc = planar_args(a, d_y, d_u, d_v, y_stride, uv_stride, ctbrows, ctbcols, ctbsize, d_bgra, pitch, n_images,
                         plane_stride_y, plane_stride_uv, image_stride, true);
    if (rc || n_images == 0) return rc;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    dim3 grid((a.width / 4 + 63) / 64, (a.height / 2 + 3) / 4, n_images);
    /* (the packed form of the 8-bit kernel was tried here too, behind a per-lane test that every sample lies in [0, 8191]: 0.778 against 0.792 of the
     * HBM peak on sixteen 8K pictures -- this kernel moves twice the input bytes per instruction and was not bound by its arithmetic) */
    hipLaunchKernelGGL((k_yuv420_to_bgra<short, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}

extern "C" int ffhip_yuv400_to_bgra_16(uint8_t *d_bgra, int pitch, const int16_t *d_y, int y_stride, int ctbrows,
                                       int ctbcols, int ctbsize, int n_images, int64_t plane_stride_y,
                                       int64_t image_stride, void *stream)
{
    PlanarArgs a;
    int rc = planar_args(a, d_y, nullptr, nullptr, y_stride, 0, ctbrows, ctbcols, ctbsize, d_bgra, pitch, n_images,
                         plane_stride_y, 0, image_stride, false);
    if (rc || n_images == 0) return rc;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    dim3 grid((a.width / 4 + 63) / 64, (a.height + 3) / 4, n_images);
    hipLaunchKernelGGL(k_yuv400_to_bgra_16, grid, dim3(256), 0, (hipStream_t)stream, a);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
