/*
 * tests/tools/check_color_int.c -- exhaustive proof-by-enumeration that the integer
 * colour forms used by the fused JPEG kernel equal the reference's double
 * arithmetic (utils/colorspace.c:162-164) on the kernel's whole input domain.
 *
 * Domain: idct_8x8_16 stores clamp(v >> 18, 0, 65535) with v a 32-bit int, so
 * every sample is in [0, 8191]; hence yy in [0, 8191] and uu, vv = sample - 128
 * in [-128, 8063] (no int16 wrap).
 *
 *   R = clamp255(yy + floor(32*vv/25))
 *   B = clamp255(yy + floor(266*uu/125))
 *   G = clamp255(yy + floor(-(215*uu + 381*vv)/1000))     unless "sensitive"
 *   sensitive  <=>  (215*uu + 381*vv) % 1000 == 0 and (uu, vv) != (0, 0):
 *                   the exact value is an integer and the double roundings decide;
 *                   the kernel evaluates those pixels in fp64.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off check_color_int.c -o check_color_int
 * Usage: check_color_int [quick]   (quick: subsample uu for the G sweep)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int fdiv(int a, int b) { int q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }

int main(int argc, char **argv)
{
    int quick = argc > 1 && !strcmp(argv[1], "quick");
    long bad_r = 0, bad_b = 0, bad_g = 0, n_sens = 0, sens_diff = 0;
#pragma omp parallel for reduction(+ : bad_r, bad_b)
    for (int c = -128; c <= 8063; c++) {
        int fr = fdiv(32 * c, 25), fb = fdiv(266 * c, 125);
        for (int yy = 0; yy <= 8191; yy++) {
            int16_t y16 = (int16_t)yy, c16 = (int16_t)c;
            int r = clamp255((int)(y16 + 1.280 * c16));
            int b = clamp255((int)(y16 + 2.128 * c16));
            bad_r += r != clamp255(yy + fr);
            bad_b += b != clamp255(yy + fb);
        }
    }
    printf("R mismatches %ld, B mismatches %ld (of %ld each)\n", bad_r, bad_b, 8192L * 8192L);
    int ustep = quick ? 37 : 1;
#pragma omp parallel for reduction(+ : bad_g, n_sens, sens_diff) schedule(dynamic, 16)
    for (int uu = -128; uu <= 8063; uu += ustep) {
        for (int vv = -128; vv <= 8063; vv++) {
            int s = 215 * uu + 381 * vv;
            int fg = fdiv(-s, 1000);
            int sens = (s % 1000 == 0) && (uu != 0 || vv != 0);
            /* only yy whose exact result is within [-2, 257] can be affected by
             * rounding; outside, both forms clamp identically (checked at edges) */
            int lo = -2 - fg, hi = 257 - fg;
            if (lo < 0) lo = 0;
            if (hi > 8191) hi = 8191;
            int16_t u16 = (int16_t)uu, v16 = (int16_t)vv;
            for (int yy = lo; yy <= hi; yy++) {
                int16_t y16 = (int16_t)yy;
                int g = clamp255((int)(y16 - 0.215 * u16 - 0.381 * v16));
                int gi = clamp255(yy + fg);
                if (sens) { n_sens++; sens_diff += g != gi; }
                else bad_g += g != gi;
            }
            /* far ends */
            for (int k = 0; k < 2; k++) {
                int yy = k ? 8191 : 0;
                int16_t y16 = (int16_t)yy;
                int g = clamp255((int)(y16 - 0.215 * u16 - 0.381 * v16));
                if (!sens) bad_g += g != clamp255(yy + fg);
            }
        }
    }
    printf("G mismatches on non-sensitive chroma: %ld; sensitive (yy,uu,vv) triples %ld of which %ld differ from the integer form\n",
           bad_g, n_sens, sens_diff);
    return (bad_r || bad_b || bad_g) ? 1 : 0;
}
