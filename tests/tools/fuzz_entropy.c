/*
 * tests/tools/fuzz_entropy.c -- robustness driver for the C host-side JPEG front end
 * (ffpic_amd/csrc/ffhip_entropy.c), built with -fsanitize=address,undefined by
 * tests/test_entropy.py.  Feeds seeded corruptions (bit flips, truncations, spliced segments)
 * of the given files through ffhip_jpeg_probe / ffhip_jpeg_entropy_decode; any memory error
 * or undefined behaviour aborts the process.  No GPU involved.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ffpic_hip.h"

static unsigned long long s_rng = 88172645463325252ULL;
static unsigned rnd(void) { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return (unsigned)(s_rng >> 11); }

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 200, ok = 0, rejected = 0;
    for (int a = 2; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 2;
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        unsigned char *orig = malloc((size_t)n), *buf = malloc((size_t)n);
        if (fread(orig, 1, (size_t)n, f) != (size_t)n) return 2;
        fclose(f);
        ffhip_jpeg_geom g0;
        int w, h;
        if (ffhip_jpeg_probe(orig, (size_t)n, &g0, &w, &h)) return 3; /* the pristine file must parse */
        const size_t mcus = (size_t)g0.mcu_cols * g0.mcu_rows;
        int16_t *cy = malloc(mcus * g0.h * g0.v * 128), *cu = malloc(mcus * 128), *cv = malloc(mcus * 128);
        uint16_t q[256];
        for (int it = 0; it < iters; it++) {
            memcpy(buf, orig, (size_t)n);
            size_t len = (size_t)n;
            const int kind = it % 5;
            if (kind == 0) for (int k = 0; k < 1 + (int)(rnd() % 8); k++) buf[rnd() % n] ^= (unsigned char)(1u << (rnd() % 8));
            else if (kind == 1) len = rnd() % (unsigned)n;                               /* truncation */
            else if (kind == 2) { size_t p = rnd() % n, l = rnd() % 64; if (p + l < (size_t)n) memset(buf + p, 0xFF, l); }
            else if (kind == 3) { size_t p = 2 + rnd() % 600; if (p < (size_t)n) buf[p] = (unsigned char)rnd(); } /* header bytes */
            else { /* the 16 code-length counts of some DHT: over- and under-subscribed canonical codes */
                size_t hits[16], nh = 0;
                for (size_t p = 2; p + 21 < (size_t)n && nh < 16; p++)
                    if (buf[p] == 0xFF && buf[p + 1] == 0xC4) hits[nh++] = p;
                if (nh) {
                    const size_t p = hits[rnd() % nh] + 5; /* marker, length, Tc/Th, then the counts */
                    for (int k = 0; k < 1 + (int)(rnd() % 3); k++) buf[p + rnd() % 16] = (unsigned char)(rnd() % 4 ? rnd() % 8 : rnd());
                }
            }
            ffhip_jpeg_geom g;
            int rc = ffhip_jpeg_probe(buf, len, &g, &w, &h);
            /* decode against the ORIGINAL geometry: a corrupted header that changes the geometry must be refused */
            if (rc == 0) rc = ffhip_jpeg_entropy_decode(buf, len, &g0, cy, g0.ncomp == 3 ? cu : NULL, g0.ncomp == 3 ? cv : NULL, q);
            if (rc == 0) ok++; else rejected++;
        }
        free(orig); free(buf); free(cy); free(cu); free(cv);
    }
    printf("decoded %d, rejected %d\n", ok, rejected);
    return 0;
}
