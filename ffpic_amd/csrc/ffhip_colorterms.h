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
#endif
