/*
 * oracle/ffo_jpeg.c -- CPU restatement of the JPEG post-entropy path.
 * TEST INFRASTRUCTURE ONLY (see oracle/ffo.h).
 *
 * Follows (reference file:line, /root/reference):
 *   ffo_jpeg_dequant          format/jpg.c:247-253   dequant_data_unit
 *   ffo_idct_8x8_16           utils/idct.c:512-534   idct_8x8_16
 *                             utils/idct.c:380-387   idct_1d_8_16bit
 *                             utils/idct.c:358-367   idct_transform_p13
 *   ffo_yuv_to_bgra32_mcu16   utils/colorspace.c:133-172 YUV_to_BGRA32_16bit
 *   ffo_jpeg_recon_image      format/jpg.c:512-560   MCU loop of JPG_decode_scan
 *
 * All integer sums are carried in uint32_t so the two's-complement wrap the
 * reference gets from -fwrapv is defined behaviour here.
 */
#include "ffo.h"

#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* 13-bit fixed point basis, rows 0..3; rows 7-i are (-1)^u mirrored
 * (utils/idct.c:358-367).  Even columns (u = 0,2,4,6) and odd columns
 * (u = 1,3,5,7) are listed separately because the +-1 "libjpeg mimic" tweaks
 * make them irregular (10703 vs 10704, 2259/2260/2261, 6436/6437). */
static const int32_t k_even[4][4] = {
    {8192, 10703, 8192, 4433},
    {8192, 4433, -8192, -10704},
    {8192, -4433, -8192, 10704},
    {8192, -10703, 8192, -4433},
};
static const int32_t k_odd[4][4] = {
    {11363, 9633, 6437, 2260},
    {9633, -2259, -11362, -6436},
    {6437, -11362, 2261, 9633},
    {2260, -6436, 9633, -11363},
};

static int32_t basis(int i, int u)
{
    int m = i < 4 ? i : 7 - i;
    int32_t c = (u & 1) ? k_odd[m][u >> 1] : k_even[m][u >> 1];
    return (i >= 4 && (u & 1)) ? -c : c;
}

static int32_t asr(uint32_t x, int s)
{
    /* arithmetic shift right of a two's complement value held in a uint32_t */
    return (x & 0x80000000u) ? (int32_t)~((~x) >> s) : (int32_t)(x >> s);
}

/* out[i] = sum_u T[8i+u] * in[u*stride]   (idct.c:380-387), mod 2^32 */
static void idct_1d_8(const int16_t *in, int stride, uint32_t out[8])
{
    for (int i = 0; i < 8; i++) {
        uint32_t acc = 0;
        for (int u = 0; u < 8; u++)
            acc += (uint32_t)basis(i, u) * (uint32_t)(int32_t)in[u * stride];
        out[i] = acc;
    }
}

void ffo_jpeg_dequant(int16_t dst[64], const int16_t src[64], const uint16_t quant[64], int end)
{
    /* jpg.c:250-252: int product, modular store to int16 */
    for (int i = 0; i <= end; i++)
        dst[i] = (int16_t)(uint16_t)((uint32_t)(int32_t)src[i] * (uint32_t)quant[i]);
}

void ffo_idct_8x8_16(int16_t blk[64])
{
    int16_t col[64];
    uint32_t buf[8];
    /* pass 1: columns, (sum + 2^10) >> 11, modular store to int16 (idct.c:517-523) */
    for (int x = 0; x < 8; x++) {
        idct_1d_8(blk + x, 8, buf);
        for (int y = 0; y < 8; y++)
            col[8 * y + x] = (int16_t)(uint16_t)asr(buf[y] + 1024u, 11);
    }
    /* pass 2: rows, (sum + 257*2^17) >> 18, clamp to [0,65535], modular store
     * (idct.c:524-532).  257<<17 = level shift 128 + rounding 0.5. */
    for (int y = 0; y < 8; y++) {
        idct_1d_8(col + 8 * y, 1, buf);
        for (int x = 0; x < 8; x++) {
            int32_t v = asr(buf[x] + (257u << 17), 18);
            v = v < 0 ? 0 : (v > 65535 ? 65535 : v);
            blk[8 * y + x] = (int16_t)(uint16_t)v;
        }
    }
}

/* clamp(int v, int M) of utils/utils.h:41-44 applied to a double argument: the
 * implicit double->int conversion truncates toward zero first. */
static inline int clamp_trunc(double d, int M)
{
    int v = (int)d;
    return v < 0 ? 0 : (v > M ? M : v);
}

void ffo_yuv_to_bgra32_mcu16(uint8_t *dst, int pitch, const int16_t *Y, const int16_t *U,
                             const int16_t *V, int v, int h)
{
    for (int i = 0; i < 8 * v; i++) {
        uint8_t *p = dst + (int64_t)i * pitch;
        for (int k = 0; k < 8 * h; k++) {
            /* luma blocks are stored one 8x8 after another, order vi*h+hi (colorspace.c:148) */
            int16_t yy = Y[((i / 8) * h + (k / 8)) * 64 + (i % 8) * 8 + (k % 8)];
            /* nearest-neighbour chroma; the -128 result is stored to int16 (colorspace.c:149-150) */
            int16_t uu = (int16_t)(uint16_t)(U[(i / v) * 8 + (k / h)] - 128);
            int16_t vv = (int16_t)(uint16_t)(V[(i / v) * 8 + (k / h)] - 128);
            /* literal double expressions, left-to-right, no FMA (build: -ffp-contract=off) */
            int r = clamp_trunc(yy + 1.280 * vv, 255);
            int g = clamp_trunc(yy - 0.215 * uu - 0.381 * vv, 255);
            int b = clamp_trunc(yy + 2.128 * uu, 255);
            p[4 * k + 0] = (uint8_t)b;
            p[4 * k + 1] = (uint8_t)g;
            p[4 * k + 2] = (uint8_t)r;
            p[4 * k + 3] = 0xff;
        }
    }
}

static int geom_ok(const ffo_jpeg_geom *g)
{
    if (!g || g->mcu_cols <= 0 || g->mcu_rows <= 0) return 0;
    if (g->ncomp != 1 && g->ncomp != 3) return 0;
    /* the MCU scratch is Y[3][64*4] (jpg.c:501): any sampling pair with h*v <= 4 data units, i.e. also
     * 4:1:1 (h = 4) and its transpose (v = 4) and the three-block pairs; YUV_to_BGRA32_16bit itself takes
     * any (v, h) (colorspace.c:143-150) */
    if (g->h < 1 || g->v < 1 || g->h * g->v > 4) return 0;
    for (int c = 0; c < g->ncomp; c++)
        if (g->qt_id[c] < 0 || g->qt_id[c] > 3) return 0;
    return 1;
}

int ffo_jpeg_recon_image(const ffo_jpeg_geom *g, const int16_t *coef_y, const int16_t *coef_u,
                         const int16_t *coef_v, const uint16_t quant[4][64], uint8_t *bgra,
                         int64_t pitch)
{
    if (!geom_ok(g)) return -EINVAL;
    const int h = g->h, v = g->v, nb = h * v;
    int16_t Y[3][64 * 4];
    /* grey: U = V = a block of zeros, never transformed (jpg.c:501,552-554) */
    static const int16_t dummy[64] = {0};
    for (int my = 0; my < g->mcu_rows; my++) {
        for (int mx = 0; mx < g->mcu_cols; mx++) {
            int64_t mcu = (int64_t)my * g->mcu_cols + mx;
            /* dequant + IDCT of every data unit of the MCU (jpg.c:540-550) */
            for (int b = 0; b < nb; b++) {
                ffo_jpeg_dequant(&Y[0][64 * b], coef_y + (mcu * nb + b) * 64, quant[g->qt_id[0]], 63);
                ffo_idct_8x8_16(&Y[0][64 * b]);
            }
            if (g->ncomp == 3) {
                ffo_jpeg_dequant(Y[1], coef_u + mcu * 64, quant[g->qt_id[1]], 63);
                ffo_idct_8x8_16(Y[1]);
                ffo_jpeg_dequant(Y[2], coef_v + mcu * 64, quant[g->qt_id[2]], 63);
                ffo_idct_8x8_16(Y[2]);
            }
            uint8_t *ptr = bgra + (int64_t)my * 8 * v * pitch + (int64_t)mx * 8 * h * 4;
            ffo_yuv_to_bgra32_mcu16(ptr, (int)pitch, Y[0], g->ncomp == 3 ? Y[1] : dummy,
                                    g->ncomp == 3 ? Y[2] : dummy, v, h);
        }
    }
    return 0;
}

struct batch_job {
    const ffo_jpeg_geom *g;
    int first, last;
    const int16_t *cy, *cu, *cv;
    const uint16_t *quant;
    int64_t quant_stride;
    uint8_t *bgra;
    int64_t pitch, image_stride;
    int rc;
};

static void *batch_worker(void *arg)
{
    struct batch_job *j = (struct batch_job *)arg;
    const ffo_jpeg_geom *g = j->g;
    int64_t mcus = (int64_t)g->mcu_cols * g->mcu_rows;
    int64_t ylen = mcus * g->h * g->v * 64, clen = mcus * 64;
    for (int i = j->first; i < j->last; i++) {
        const uint16_t(*q)[64] = (const uint16_t(*)[64])(j->quant + (int64_t)i * j->quant_stride);
        int rc = ffo_jpeg_recon_image(g, j->cy + i * ylen, j->cu ? j->cu + i * clen : NULL,
                                      j->cv ? j->cv + i * clen : NULL, q,
                                      j->bgra + (int64_t)i * j->image_stride, j->pitch);
        if (rc) j->rc = rc;
    }
    return NULL;
}

int ffo_jpeg_recon_batch(const ffo_jpeg_geom *g, int n_images, const int16_t *coef_y,
                         const int16_t *coef_u, const int16_t *coef_v, const uint16_t *quant,
                         int64_t quant_stride, uint8_t *bgra, int64_t pitch, int64_t image_stride,
                         int n_threads)
{
    if (!geom_ok(g) || n_images < 0) return -EINVAL;
    if (n_images == 0) return 0;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_images) n_threads = n_images;
    struct batch_job *jobs = calloc((size_t)n_threads, sizeof *jobs);
    pthread_t *tid = calloc((size_t)n_threads, sizeof *tid);
    if (!jobs || !tid) { free(jobs); free(tid); return -ENOMEM; }
    int rc = 0;
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (struct batch_job){g, (int)((int64_t)n_images * t / n_threads),
                                     (int)((int64_t)n_images * (t + 1) / n_threads),
                                     coef_y, coef_u, coef_v, quant, quant_stride,
                                     bgra, pitch, image_stride, 0};
        if (t > 0) pthread_create(&tid[t], NULL, batch_worker, &jobs[t]);
    }
    batch_worker(&jobs[0]);
    for (int t = 1; t < n_threads; t++) pthread_join(tid[t], NULL);
    for (int t = 0; t < n_threads; t++)
        if (jobs[t].rc) rc = jobs[t].rc;
    free(jobs);
    free(tid);
    return rc;
}
