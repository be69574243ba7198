/*
 * transbmp_hip.c -- the reference's app/transbmp.c (file -> BMP) done through the C ABI of
 * libffpic_hip.so only: host-side entropy front end, fused reconstruction on the MI355X, BMP
 * sink.  Plain C11; no HIP headers, no C++.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/transbmp_hip.c -Lffpic_amd -lffpic_hip \
 *       -Wl,-rpath,$PWD/ffpic_amd -o transbmp_hip
 *   ./transbmp_hip picture.jpg [more.jpg ...]       # writes "picture.jpg (W * H).bmp" like transbmp
 *
 * Pictures of the same geometry given on one command line are reconstructed as one batch.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ffpic_hip.h"

static unsigned char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *b = malloc((size_t)n + 1);
    if (b && fread(b, 1, (size_t)n, f) != (size_t)n) { free(b); b = NULL; }
    fclose(f);
    *len = (size_t)n;
    return b;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.jpg [...]\n", argv[0]); return 2; }
    const int n = argc - 1;
    int rc = ffhip_init(0);
    if (rc) { fprintf(stderr, "no gfx950 device: %s (use the reference's C path)\n", ffhip_strerror(rc)); return 1; }

    const unsigned char **files = calloc((size_t)n, sizeof *files);
    size_t *lens = calloc((size_t)n, sizeof *lens);
    int *status = calloc((size_t)n, sizeof *status), w0 = 0, h0 = 0;
    ffhip_jpeg_geom g;
    for (int i = 0; i < n; i++) {
        files[i] = slurp(argv[1 + i], &lens[i]);
        ffhip_jpeg_geom gi;
        int w, h;
        if (!files[i] || (rc = ffhip_jpeg_probe(files[i], lens[i], &gi, &w, &h))) {
            fprintf(stderr, "%s: not a baseline JPEG this back end decodes (%d)\n", argv[1 + i], rc);
            return 1;
        }
        if (i == 0) { g = gi; w0 = w; h0 = h; }
        else if (memcmp(&g, &gi, sizeof g)) { fprintf(stderr, "%s: geometry differs from the first file\n", argv[1 + i]); return 1; }
    }
    const size_t mcus = (size_t)g.mcu_cols * g.mcu_rows, yb = mcus * g.h * g.v * 64, cb = mcus * 64;
    const int W = g.mcu_cols * 8 * g.h, H = g.mcu_rows * 8 * g.v;
    int16_t *cy = malloc(n * yb * 2), *cu = malloc(n * cb * 2), *cv = malloc(n * cb * 2);
    uint16_t *quant = malloc((size_t)n * 256 * 2);
    uint8_t *bgra = malloc((size_t)n * W * H * 4);
    if (!cy || !cu || !cv || !quant || !bgra) return 1;

    rc = ffhip_jpeg_entropy_batch(files, lens, n, 8, &g, cy, g.ncomp == 3 ? cu : NULL, g.ncomp == 3 ? cv : NULL, quant, status);
    if (rc) { fprintf(stderr, "entropy decode failed: %d\n", rc); return 1; }
    rc = ffhip_jpeg_recon_batch_host(&g, n, cy, g.ncomp == 3 ? cu : NULL, g.ncomp == 3 ? cv : NULL, quant, 256, bgra,
                                     (int64_t)W * 4, (int64_t)W * H * 4);
    if (rc) { fprintf(stderr, "reconstruction failed: %s\n", ffhip_strerror(rc)); return 1; }
    for (int i = 0; i < n; i++) {
        char name[1024];
        /* the reference names its output "<file> (<w> * <h>).bmp" with w aligned to 8 (app/transbmp.c, format/jpg.c:794) */
        const int wa = (w0 + 7) & ~7;
        snprintf(name, sizeof name, "%s (%d * %d).bmp", argv[1 + i], wa, h0);
        rc = ffhip_bmp_write(name, bgra + (size_t)i * W * H * 4, wa, h0, (int64_t)W * 4);
        if (rc) { fprintf(stderr, "cannot write %s\n", name); return 1; }
        printf("%s\n", name);
    }
    return 0;
}
