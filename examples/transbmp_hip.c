/*
 * transbmp_hip.c -- the reference's app/transbmp.c (file -> BMP) done through the C ABI of
 * libffpic_hip.so only: ffhip_jpeg_decode_files (host-side entropy front end overlapped with the fused
 * reconstruction on the MI355X), then the BMP sink.  Plain C11; no HIP headers, no C++.
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
    const int W = g.mcu_cols * 8 * g.h, H = g.mcu_rows * 8 * g.v;
    uint8_t *bgra = ffhip_host_malloc((size_t)n * W * H * 4); /* pinned: the device copies straight into it */
    if (!bgra) return 1;
    /* Huffman decode on 8 host threads, overlapped chunk by chunk with copy + reconstruction on the GPU */
    rc = ffhip_jpeg_decode_files(files, lens, n, 8, 0, &g, bgra, (int64_t)W * 4, (int64_t)W * H * 4, status);
    if (rc) { fprintf(stderr, "decode failed: %s\n", ffhip_strerror(rc)); return 1; }
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
