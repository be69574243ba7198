/*
 * tests/tools/check_color_fma.c -- exhaustive check of the float forms of the chroma terms the JPEG and planar colour
 * kernels compute (ffhip_colorterms.h::ff_chroma_terms): one v_fma_f32 and one v_add_f32 per term, the add onto
 * 2^23 + 65536 - bias leaving the term's two's-complement int16 in the low half of the float's bit pattern,
 * from the RAW chroma samples a = uu + 128, b = vv + 128 in [0, 8191] (af, bf their float values):
 *     fr = floor(32 vv / 25)                  = low16(bits(fma(bf, 1.28, -0.32)  + 8453980))
 *     fb = floor(266 uu / 125)                = low16(bits(fma(af, 2.128, 0.12)  + 8453871))
 *     fg = floor(-(215 uu + 381 vv) / 1000)   = low16(bits(fma(tf, 0.001, -0.4995) + 8449338)),  tf = 4882288 - (215 af + 381 bf)
 *     sensitive <=> 215 uu + 381 vv is a non-zero multiple of 1000  <=> fma(kf, -1000, tf) == 0 and 215 af + 381 bf != 76288
 * against the integer definitions, for every a, b in [0, 8191] (the IDCT output domain).
 * The (x + 0.5)/d - 0.5 offsets keep every exact value at least 0.5/d away from a rounding boundary; the enumeration
 * shows the float32 errors stay inside that margin.  fmaf is correctly rounded (the same single rounding as v_fma_f32).
 *
 * Build: gcc -O2 -fopenmp -mfma -ffp-contract=off check_color_fma.c -o check_color_fma -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline int fdiv(int a, int b) { int q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }
static inline int low16(float f) { uint32_t u; memcpy(&u, &f, 4); return (int16_t)(u & 0xffffu); }

int main(void)
{
    long bad_r = 0, bad_b = 0, bad_g = 0, bad_s = 0, n_sens = 0;
    for (int c = -128; c <= 8063; c++) {
        const float cf = (float)(c + 128);
        volatile float tr = fmaf(cf, 1.28f, -0.32f);
        volatile float tb = fmaf(cf, 2.128f, 0.12f);
        bad_r += low16(tr + 8453980.0f) != fdiv(32 * c, 25);
        bad_b += low16(tb + 8453871.0f) != fdiv(266 * c, 125);
    }
#pragma omp parallel for reduction(+ : bad_g, bad_s, n_sens) schedule(static)
    for (int uu = -128; uu <= 8063; uu++) {
        const float uf = (float)(uu + 128);
        for (int vv = -128; vv <= 8063; vv++) {
            const float vf = (float)(vv + 128);
            const float sf = fmaf(vf, 381.0f, uf * 215.0f);
            const float tf = 4882288.0f - sf;
            const float q = fmaf(tf, 0.001f, -0.4995f);
            const float tg = q + 8449338.0f;
            const float kf = tg - 8449338.0f;
            const float rem = fmaf(kf, -1000.0f, tf);
            const int s = 215 * uu + 381 * vv;
            const int sens = (s % 1000 == 0) && s != 0;
            bad_g += low16(tg) != fdiv(-s, 1000) || (int)kf != fdiv(4806000 - s, 1000) || (int)sf != s + 76288;
            bad_s += ((rem == 0.0f) && (sf != 76288.0f)) != sens;
            n_sens += sens;
        }
    }
    printf("fr mismatches %ld, fb mismatches %ld (8192 each); fg mismatches %ld, sensitivity mismatches %ld (of %ld pairs, %ld sensitive)\n",
           bad_r, bad_b, bad_g, bad_s, 8192L * 8192L, n_sens);
    return (bad_r || bad_b || bad_g || bad_s) ? 1 : 0;
}
