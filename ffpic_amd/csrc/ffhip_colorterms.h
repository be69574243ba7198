/* ffhip_colorterms.h -- the reference's YCbCr->RGB arithmetic (utils/colorspace.c:162-164,
 * 316-318, 653-655) in the two forms the kernels use:
 *   - literal fp64 (every operation rounded separately; compile with -ffp-contract=off);
 *   - exact integer forms valid for yy in [0,8191], uu,vv in [-128,8063]
 *     (tests/tools/check_color_int.c, tests/test_color_forms.py):
 *        R = clamp255(yy + floor(32 vv/25)),  B = clamp255(yy + floor(266 uu/125)),
 *        G = clamp255(yy + floor(-(215 uu + 381 vv)/1000))  unless 215 uu + 381 vv is a
 *        non-zero multiple of 1000 ("sensitive": the double roundings decide).
 */
#ifndef FFHIP_COLORTERMS_H
#define FFHIP_COLORTERMS_H
#include "ffhip_internal.h"

__device__ __forceinline__ int ff_clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* exact floor(x/d), x >= 0: (int)((float)(2x+1) * fl(1/(2d))), enumerated for d = 25, 125, 1000 */
__device__ __forceinline__ int ff_fdiv(int two_x_plus_1, float inv_2d) { return (int)((float)two_x_plus_1 * inv_2d); }

struct ChromaTerms {
    int fr, fg, fb;
    bool sensitive;
};

__device__ __forceinline__ ChromaTerms ff_chroma_terms(int uu, int vv)
{
    ChromaTerms t;
    t.fr = ff_fdiv(64 * vv + (2 * 25 * 164 + 1), 1.0f / 50.0f) - 164;
    t.fb = ff_fdiv(532 * uu + (2 * 125 * 273 + 1), 1.0f / 250.0f) - 273;
    const int sgm = 215 * uu + 381 * vv;
    const int n = 4806000 - sgm;
    const int nq = ff_fdiv(2 * n + 1, 1.0f / 2000.0f);
    t.fg = nq - 4806;
    t.sensitive = (n - nq * 1000 == 0) && sgm != 0;
    return t;
}

/* literal double arithmetic; yy, uu, vv are the int16 values the reference holds */
__device__ __forceinline__ u32 ff_bgra_fp64(int yy, int uu, int vv)
{
    const double dr = (double)yy + 1.280 * (double)vv;
    double dg = (double)yy - 0.215 * (double)uu;
    dg = dg - 0.381 * (double)vv;
    const double db = (double)yy + 2.128 * (double)uu;
    return (u32)ff_clamp255((int)db) | ((u32)ff_clamp255((int)dg) << 8) | ((u32)ff_clamp255((int)dr) << 16) | 0xff000000u;
}

__device__ __forceinline__ u32 ff_green_fp64(int yy, int uu, int vv)
{
    double g = (double)yy - 0.215 * (double)uu;
    g = g - 0.381 * (double)vv;
    return (u32)ff_clamp255((int)g);
}

/* one pixel through the integer forms (caller guarantees the domain and !sensitive) */
__device__ __forceinline__ u32 ff_bgra_int(int yy, const ChromaTerms &t)
{
    return (u32)ff_clamp255(yy + t.fb) | ((u32)ff_clamp255(yy + t.fg) << 8) | ((u32)ff_clamp255(yy + t.fr) << 16) | 0xff000000u;
}

/* ---- the packed forms of the fused kernels (ffhip_jpeg.hip, ffhip_vp8_frame.hip): two pixels per dword ---- */
__device__ __forceinline__ u32 sat_pk_u8_i16(u32 v)
{
    u32 d;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(v));
    return d;
}

__device__ __forceinline__ u32 pk_add16(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}

/* literal fp64 G of colorspace.c:163 (contraction is off for this file) */
__device__ __forceinline__ u32 green_fp64(int yy, int uu, int vv)
{
    double g = (double)yy - 0.215 * (double)uu;
    g = g - 0.381 * (double)vv;
    int gi = (int)g;
    return (u32)(gi < 0 ? 0 : (gi > 255 ? 255 : gi));
}

/* The same terms for TWO chroma samples at once, in the packed fp32 forms (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: one issue
 * for both samples): uw, vw hold two raw 16-bit samples each.  r, g, b come back as the two int16 terms side by side (sample 0
 * low), ready for the packed 16-bit adds.  The "sensitive" test is left to the caller in two steps: rem (= tf mod 1000, exact)
 * is zero where 215 uu + 381 vv is a multiple of 1000 -- a product over the samples of a pass says whether ANY is -- and only
 * then is sf != 76288 (the multiple is not zero itself) looked at, per sample: 3 instructions per pass in the common case where
 * the two compares, the and and the mask insertion per SAMPLE used to be. */
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct TermBits2 {
    u32 r, g, b;
    f32x2 rem, sf;
};
__device__ __forceinline__ TermBits2 chroma_term_bits2(u32 uw, u32 vw)
{
    f32x2 af, bf; /* the halfword select rides on the conversion (SDWA) */
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(af.x) : "v"(uw));
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(af.y) : "v"(uw));
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(bf.x) : "v"(vw));
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(bf.y) : "v"(vw));
    const f32x2 r = __builtin_elementwise_fma(bf, (f32x2)(1.28f), (f32x2)(-0.32f)) + (f32x2)(8453980.0f);
    const f32x2 b = __builtin_elementwise_fma(af, (f32x2)(2.128f), (f32x2)(0.12f)) + (f32x2)(8453871.0f);
    TermBits2 t;
    t.sf = __builtin_elementwise_fma(bf, (f32x2)(381.0f), af * (f32x2)(215.0f));
    const f32x2 tf = (f32x2)(4882288.0f) - t.sf;
    const f32x2 tg = __builtin_elementwise_fma(tf, (f32x2)(0.001f), (f32x2)(-0.4995f)) + (f32x2)(8449338.0f);
    const f32x2 kf = tg - (f32x2)(8449338.0f);
    t.rem = __builtin_elementwise_fma(kf, (f32x2)(-1000.0f), tf);
    /* (whole-vector bit casts: __builtin_bit_cast of ONE element of an ext_vector reads element 0 whichever is named, clang 20) */
    const u32x2 rb = __builtin_bit_cast(u32x2, r), gb = __builtin_bit_cast(u32x2, tg), bb = __builtin_bit_cast(u32x2, b);
    t.r = __builtin_amdgcn_perm(rb[1], rb[0], 0x05040100u);
    t.g = __builtin_amdgcn_perm(gb[1], gb[0], 0x05040100u);
    t.b = __builtin_amdgcn_perm(bb[1], bb[0], 0x05040100u);
    return t;
}
/* Four pixels of one row at 4:2:0 in the packed forms: `el` four 8-bit luma samples, `eu` / `ev` the two 8-bit chroma samples they share
 * (pixels 0, 1 the first, 2, 3 the second).  The chroma terms are made once (ff_packed420_terms) and serve every row that shares the
 * samples; a row is three 16-bit adds, three saturating packs and three byte permutes per pixel PAIR, fp64 only where 215 uu + 381 vv is a
 * non-zero multiple of 1000.  Exact for 8-bit samples (the forms hold on [0, 8191]: tests/tools/check_color_fma.c). */
struct Packed420 {
    u32 tr2[2], tg2[2], tb2[2];
    f32x2 rem, sf;
    u32 eu, ev;
};
__device__ __forceinline__ Packed420 ff_packed420_terms(const u32 eu, const u32 ev)
{
    const u32 uw = __builtin_amdgcn_perm(0u, eu, 0x0c010c00u), vw = __builtin_amdgcn_perm(0u, ev, 0x0c010c00u); /* two raw samples as halfwords */
    const TermBits2 t = chroma_term_bits2(uw, vw);
    Packed420 p;
    p.tr2[0] = __builtin_amdgcn_perm(t.r, t.r, 0x01000100u); p.tr2[1] = __builtin_amdgcn_perm(t.r, t.r, 0x03020302u);
    p.tg2[0] = __builtin_amdgcn_perm(t.g, t.g, 0x01000100u); p.tg2[1] = __builtin_amdgcn_perm(t.g, t.g, 0x03020302u);
    p.tb2[0] = __builtin_amdgcn_perm(t.b, t.b, 0x01000100u); p.tb2[1] = __builtin_amdgcn_perm(t.b, t.b, 0x03020302u);
    p.rem = t.rem; p.sf = t.sf; p.eu = eu; p.ev = ev;
    return p;
}
__device__ __forceinline__ u32x4 ff_packed420_row(const Packed420 &p, const u32 el)
{
    const u32 yp[2] = {__builtin_amdgcn_perm(0u, el, 0x0c010c00u), __builtin_amdgcn_perm(0u, el, 0x0c030c02u)};
    u32x4 px;
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
        const u32 r2 = sat_pk_u8_i16(pk_add16(yp[h2], p.tr2[h2]));
        const u32 g2 = sat_pk_u8_i16(pk_add16(yp[h2], p.tg2[h2]));
        const u32 b2 = sat_pk_u8_i16(pk_add16(yp[h2], p.tb2[h2]));
        const u32 bg = __builtin_amdgcn_perm(g2, b2, 0x05010400u); /* b0 g0 b1 g1 */
        px[2 * h2] = __builtin_amdgcn_perm(r2, bg, 0x0d040100u);     /* b0 g0 r0 ff */
        px[2 * h2 + 1] = __builtin_amdgcn_perm(r2, bg, 0x0d050302u); /* b1 g1 r1 ff */
    }
    if (p.rem.x * p.rem.y == 0.0f) { /* rare: some sample's G sum is a multiple of 1000 (zero included) */
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int k2 = d >> 1;
            const float rem = k2 ? p.rem.y : p.rem.x, sf = k2 ? p.sf.y : p.sf.x;
            if (rem == 0.0f && sf != 76288.0f) {
                const int uu = (int)((p.eu >> (8 * k2)) & 0xff) - 128, vv = (int)((p.ev >> (8 * k2)) & 0xff) - 128;
                px[d] = (px[d] & 0xffff00ffu) | (green_fp64((int)((el >> (8 * d)) & 0xff), uu, vv) << 8);
            }
        }
    }
    return px;
}

#endif
