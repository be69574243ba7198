/*
 * oracle/ref_statics_jpg.c -- part of the oracle/_ref build recipe.
 * TEST INFRASTRUCTURE ONLY; compiles only where /root/reference exists.
 *
 * Compiles the reference's own format/jpg.c (found through -I$(REF)/format,
 * never copied) inside this translation unit so that its `static` hot-path
 * function dequant_data_unit (jpg.c:247-253) can be called from tests, and adds
 * a driver that walks MCU-order coefficient planes exactly like the MCU loop
 * of JPG_decode_scan (jpg.c:512-560), calling ONLY reference code for the
 * arithmetic: dequant_data_unit, get_dct_ops(16)->idct_8x8,
 * get_cs_ops(16)->YUV_to_BGRA32.
 */
#include "jpg.c" /* the reference's format/jpg.c */

void ref_jpeg_dequant(int16_t *dst, int16_t *src, uint16_t *quant, int end)
{
    struct jpg_decoder d;
    memset(&d, 0, sizeof d);
    d.quant = quant;
    dequant_data_unit(&d, dst, src, end);
}

void ref_idct_8x8_16(int16_t *blk) { get_dct_ops(16)->idct_8x8(blk, 8); }

void ref_yuv_to_bgra32_mcu16(uint8_t *dst, int pitch, int16_t *Y, int16_t *U, int16_t *V, int v, int h)
{
    get_cs_ops(16)->YUV_to_BGRA32(dst, pitch, Y, U, V, v, h);
}

/* geom = {mcu_cols, mcu_rows, ncomp, h, v, qt_id[3]} as int32, see oracle/ffo.h */
int ref_jpeg_recon_image(const int32_t *geom, int16_t *coef_y, int16_t *coef_u, int16_t *coef_v,
                         uint16_t *quant /* [4][64] */, uint8_t *bgra, int64_t pitch)
{
    const int mcu_cols = geom[0], mcu_rows = geom[1], ncomp = geom[2], h = geom[3], v = geom[4];
    const struct dct_ops *dct = get_dct_ops(16);
    const struct cs_ops *cs_bgr = get_cs_ops(16);
    struct jpg_decoder d[3];
    int16_t Y[3][64 * 4];
    int16_t dummy[64] = {0};
    memset(d, 0, sizeof d);
    for (int c = 0; c < ncomp; c++) d[c].quant = quant + 64 * geom[5 + c];
    for (int my = 0; my < mcu_rows; my++)
        for (int mx = 0; mx < mcu_cols; mx++) {
            int64_t mcu = (int64_t)my * mcu_cols + mx;
            for (int b = 0; b < h * v; b++) {
                dequant_data_unit(&d[0], &Y[0][64 * b], coef_y + (mcu * h * v + b) * 64, 63);
                dct->idct_8x8(&Y[0][64 * b], 8);
            }
            if (ncomp == 3) {
                dequant_data_unit(&d[1], Y[1], coef_u + mcu * 64, 63);
                dct->idct_8x8(Y[1], 8);
                dequant_data_unit(&d[2], Y[2], coef_v + mcu * 64, 63);
                dct->idct_8x8(Y[2], 8);
            }
            cs_bgr->YUV_to_BGRA32(bgra + (int64_t)my * 8 * v * pitch + (int64_t)mx * 8 * h * 4,
                                  (int)pitch, Y[0], ncomp == 3 ? Y[1] : dummy,
                                  ncomp == 3 ? Y[2] : dummy, v, h);
        }
    return 0;
}
