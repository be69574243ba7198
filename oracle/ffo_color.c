/*
 * oracle/ffo_color.c -- CPU restatement of the planar YUV -> BGRA converters.
 * TEST INFRASTRUCTURE ONLY (see oracle/ffo.h).
 *
 * Follows (reference file:line, /root/reference):
 *   ffo_yuv420_to_bgra32        utils/colorspace.c:291-329 YUV420_to_BGRA32 (WebP, uint8 planes)
 *   ffo_yuv420_to_bgra32_16bit  utils/colorspace.c:628-669 YUV420_to_BGRA32_16bit (HEVC)
 *   ffo_yuv400_to_bgra32_16bit  utils/colorspace.c:715-742 YUV400_to_BGRA32_16bit
 *
 * The reference walks the picture block by block (16x16 MBs / ctbsize CTBs);
 * the pointer arithmetic reduces to plain raster addressing, which is what is
 * written here.  Build with -ffp-contract=off.
 */
#include "ffo.h"

static inline uint8_t clamp_trunc_u8(double d)
{
    int v = (int)d; /* double -> int truncates toward zero (utils.h:41-44) */
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

static inline void put_px(uint8_t *p, int16_t yy, int16_t u, int16_t v)
{
    p[0] = clamp_trunc_u8(yy + 2.128 * u);
    p[1] = clamp_trunc_u8(yy - 0.215 * u - 0.381 * v);
    p[2] = clamp_trunc_u8(yy + 1.280 * v);
    p[3] = 0xff;
}

void ffo_yuv420_to_bgra32(uint8_t *dst, int pitch, const uint8_t *y, const uint8_t *u,
                          const uint8_t *v, int y_stride, int uv_stride, int mbrows, int mbcols)
{
    /* MB (bx,by) has its luma origin at (y_stride*by + bx)*16 and is indexed
     * Y[i*y_stride + j] (colorspace.c:303,309): strides are in samples and the
     * walk is a plain raster; chroma sample is (r/2, c/2). */
    for (int r = 0; r < 16 * mbrows; r++)
        for (int c = 0; c < 16 * mbcols; c++) {
            int16_t yy = y[(int64_t)r * y_stride + c];
            int64_t co = (int64_t)(r / 2) * uv_stride + c / 2;
            int16_t uu = (int16_t)(u[co] - 128);
            int16_t vv = (int16_t)(v[co] - 128);
            put_px(dst + (int64_t)r * pitch + 4 * c, yy, uu, vv);
        }
}

void ffo_yuv420_to_bgra32_16bit(uint8_t *dst, int pitch, const int16_t *y, const int16_t *u,
                                const int16_t *v, int y_stride, int uv_stride, int mbrows,
                                int mbcols, int ctbsize)
{
    /* CTB (bx,by): Y origin y_stride*by*ctbsize + bx*ctbsize, chroma origin
     * (ctbsize/2)*uv_stride*by + bx*ctbsize/2 (colorspace.c:641-643); raster again
     * for even ctbsize.  u-128 / v-128 are stored to int16 (modular). */
    for (int r = 0; r < ctbsize * mbrows; r++)
        for (int c = 0; c < ctbsize * mbcols; c++) {
            int16_t yy = y[(int64_t)r * y_stride + c];
            int64_t co = (int64_t)(r / 2) * uv_stride + c / 2;
            int16_t uu = (int16_t)(uint16_t)(u[co] - 128);
            int16_t vv = (int16_t)(uint16_t)(v[co] - 128);
            put_px(dst + (int64_t)r * pitch + 4 * c, yy, uu, vv);
        }
}

void ffo_yuv400_to_bgra32_16bit(uint8_t *dst, int pitch, const int16_t *y, int y_stride,
                                int mbrows, int mbcols, int ctbsize)
{
    for (int r = 0; r < ctbsize * mbrows; r++)
        for (int c = 0; c < ctbsize * mbcols; c++) {
            int s = y[(int64_t)r * y_stride + c];
            uint8_t g = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
            uint8_t *p = dst + (int64_t)r * pitch + 4 * c;
            p[0] = p[1] = p[2] = p[3] = g; /* alpha too: colorspace.c:731-735 */
        }
}
