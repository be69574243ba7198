/*
 * oracle/ref_record_pred.c -- part of the oracle/_ref recipe.  TEST INFRASTRUCTURE ONLY.
 *
 * An interposer for the reference's exported pred_luma / pred_chrome (format/predict.c:426,590):
 * loaded with RTLD_GLOBAL before libffpic_ref.so, it records the per-macroblock arguments the
 * reference's own VP8 decoder passes (modes and the 384 residual coefficients of vp8_prerdict_mb,
 * format/webp.c:1453-1473) and forwards to the real functions.  That is how a whole-file WebP
 * fixture gets its "per-MB residual/modes dump"  without a second bool decoder.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint8_t *g_modes;    /* [n][20] */
static int16_t *g_res;      /* [n][384] */
static int g_n, g_cap;

/* the reference's own definitions, handed over by the test (dlsym on the reference's handle) */
static void (*real_luma)(int16_t *, int, uint8_t *, uint8_t *, int, int, int);
static void (*real_chroma)(int16_t *, int, uint8_t *, uint8_t *, int, int, int);
void ref_record_set_real(void *luma, void *chroma)
{
    real_luma = (void (*)(int16_t *, int, uint8_t *, uint8_t *, int, int, int))luma;
    real_chroma = (void (*)(int16_t *, int, uint8_t *, uint8_t *, int, int, int))chroma;
}

static void grow(void)
{
    if (g_n < g_cap) return;
    g_cap = g_cap ? 2 * g_cap : 4096;
    g_modes = realloc(g_modes, (size_t)g_cap * 20);
    g_res = realloc(g_res, (size_t)g_cap * 384 * 2);
}

void pred_luma(int16_t *coff, int ymode, uint8_t imodes[16], uint8_t *dst, int stride, int x, int y)
{
    grow();
    memset(g_modes + 20 * g_n, 0, 20);
    g_modes[20 * g_n] = (uint8_t)ymode;
    memcpy(g_modes + 20 * g_n + 2, imodes, 16);
    memcpy(g_res + 384 * g_n, coff, 384 * 2);
    real_luma(coff, ymode, imodes, dst, stride, x, y);
}

void pred_chrome(int16_t *coff, int imode, uint8_t *uout, uint8_t *vout, int stride, int x, int y)
{
    g_modes[20 * g_n + 1] = (uint8_t)imode; /* pred_chrome follows pred_luma of the same MB */
    g_n++;
    real_chroma(coff, imode, uout, vout, stride, x, y);
}

int ref_record_count(void) { return g_n; }
const uint8_t *ref_record_modes(void) { return g_modes; }
const int16_t *ref_record_residual(void) { return g_res; }
void ref_record_reset(void) { g_n = 0; }
