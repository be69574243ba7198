/*
 * oracle/ref_statics_webp.c -- part of the oracle/_ref build recipe.
 * TEST INFRASTRUCTURE ONLY; compiles only where /root/reference exists.
 *
 * Compiles the reference's own format/webp.c inside this translation unit and
 * exports thin wrappers around its `static` hot-path functions
 * IWHT_long / IWHT_fast (webp.c:1067-1106).
 */
#include "webp.c" /* the reference's format/webp.c */

void ref_vp8_iwht_long(const int16_t *in, int16_t *out) { IWHT_long(in, out); }
void ref_vp8_iwht_fast(int16_t *in, int16_t *out) { IWHT_fast(in, out); }
void ref_vp8_idct_4x4(int16_t *blk) { get_dct_ops(16)->idct_4x4(blk, 8); }
