/*
 * ffhip_blockops.hip -- the per-block, host-pointer, synchronous entry points that
 * stand behind the reference's own op tables:
 *   struct accl_ops  (arch/accl.h:20-25)      -> hip_accl_init / ffhip_accl_ops_get
 *   struct dct_ops   (utils/idct.h:14-21)     -> ffhip_get_dct_ops
 *   struct cs_ops    (utils/colorspace.h:29-33) -> ffhip_get_cs_ops
 *   idct_4x4_hevc    (utils/idct.h:25)        -> ffhip_idct_4x4_hevc
 *
 * They exist for drop-in compatibility and bit-exactness, not for speed: one
 * block per call means a copy in, a launch and a copy out per 32..512 bytes
 * (the reference's OpenCL back-end has the same shape, arch/opencl/opcl.c:42-88).
 * Throughput lives behind the batched API (ffhip_jpeg.hip).  The device code
 * here is deliberately a second, scalar formulation of the same arithmetic, so
 * the parity tests cross-check the packed/dot2 kernels against it as well.
 *
 * Compile with -ffp-contract=off (colour conversion is double precision).
 */
#include "ffhip_internal.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------ device code */

__device__ static const int kEven[4][4] = {
    {8192, 10703, 8192, 4433}, {8192, 4433, -8192, -10704}, {8192, -4433, -8192, 10704}, {8192, -10703, 8192, -4433}};
__device__ static const int kOdd[4][4] = {
    {11363, 9633, 6437, 2260}, {9633, -2259, -11362, -6436}, {6437, -11362, 2261, 9633}, {2260, -6436, 9633, -11363}};

/* utils/idct.c:380-387 with the table of :358-367 split into even/odd columns */
__device__ static void idct8_scalar(const short *in, int stride, u32 out[8])
{
    for (int i = 0; i < 4; i++) {
        u32 e = 0, o = 0;
        for (int k = 0; k < 4; k++) {
            e += (u32)(kEven[i][k] * (int)in[(2 * k) * stride]);
            o += (u32)(kOdd[i][k] * (int)in[(2 * k + 1) * stride]);
        }
        out[i] = e + o;
        out[7 - i] = e - o;
    }
}

/* utils/idct.c:512-534 */
__global__ void k_block_idct8x8(short *blk, int n)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    short *p = blk + b * 64, col[64];
    u32 t[8];
    for (int x = 0; x < 8; x++) {
        idct8_scalar(p + x, 8, t);
        for (int y = 0; y < 8; y++) col[8 * y + x] = (short)((int)(t[y] + 1024u) >> 11);
    }
    for (int y = 0; y < 8; y++) {
        idct8_scalar(col + 8 * y, 1, t);
        for (int x = 0; x < 8; x++) {
            int v = (int)(t[x] + (257u << 17)) >> 18;
            v = v < 0 ? 0 : (v > 65535 ? 65535 : v);
            p[8 * y + x] = (short)v;
        }
    }
}

/* VP8 4x4 inverse DCT, utils/idct.c:100-151 */
__device__ static inline int mulk(int x, int k) { return (x * k) >> 16; }
__global__ void k_block_vp8_idct4x4(short *blk, int n)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    short *p = blk + b * 16, t[16];
    for (int c = 0; c < 4; c++) {
        int x0 = p[c], x1 = p[4 + c], x2 = p[8 + c], x3 = p[12 + c];
        int s = x0 + x2, d = x0 - x2;
        int lo = mulk(x1, 35468) - x3 - mulk(x3, 20091);
        int hi = x1 + mulk(x1, 20091) + mulk(x3, 35468);
        t[c] = (short)(s + hi); t[4 + c] = (short)(d + lo); t[8 + c] = (short)(d - lo); t[12 + c] = (short)(s - hi);
    }
    for (int r = 0; r < 4; r++) {
        int x0 = t[4 * r], x1 = t[4 * r + 1], x2 = t[4 * r + 2], x3 = t[4 * r + 3];
        int s = x0 + x2, d = x0 - x2;
        int lo = mulk(x1, 35468) - x3 - mulk(x3, 20091);
        int hi = x1 + mulk(x1, 20091) + mulk(x3, 35468);
        p[4 * r] = (short)((s + hi + 4) >> 3); p[4 * r + 1] = (short)((d + lo + 4) >> 3);
        p[4 * r + 2] = (short)((d - lo + 4) >> 3); p[4 * r + 3] = (short)((s - hi + 4) >> 3);
    }
}

/* HEVC DST-VII 4x4, utils/idct.c:9-55, keeping the `+ (shift - 1)` rounding of :31 */
__device__ static void dst4_1d(const short in[4], short out[4], int cmin, int cmax, int shift)
{
    const int m[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};
    for (int i = 0; i < 4; i++) {
        u32 acc = 0;
        for (int j = 0; j < 4; j++) acc += (u32)(m[j][i] * (int)in[j]);
        int v = (int)(acc + (u32)(shift - 1)) >> shift;
        out[i] = (short)(v < cmin ? cmin : (v > cmax ? cmax : v));
    }
}
__global__ void k_block_hevc_dst4x4(const short *in, short *out, int n, int bitdepth, int epp)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const int e = epp ? (bitdepth + 6 > 15 ? bitdepth + 6 : 15) : 15;
    const int cmin = -(1 << e), cmax = (1 << e) - 1;
    int shift2 = 20 - bitdepth;
    if (epp && shift2 < 11) shift2 = 11;
    if (!epp && shift2 < 0) shift2 = 0;
    const short *p = in + b * 16;
    short *o = out + b * 16, t[4], g[4][4];
    for (int x = 0; x < 4; x++) {
        for (int y = 0; y < 4; y++) t[y] = p[x + 4 * y];
        dst4_1d(t, g[x], cmin, cmax, 7);
    }
    for (int y = 0; y < 4; y++) {
        for (int x = 0; x < 4; x++) t[x] = g[x][y];
        dst4_1d(t, o + 4 * y, cmin, cmax, shift2);
    }
}

/* utils/colorspace.c:133-172 for one MCU; output compact [8v][8h] BGRA */
__global__ void k_mcu_color(const short *Y, const short *U, const short *V, int v, int h, u32 *out)
{
    const int w = 8 * h, n = 8 * v * w;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int i = t / w, k = t - i * w;
        const short yy = Y[((i / 8) * h + (k / 8)) * 64 + (i % 8) * 8 + (k % 8)];
        const short uu = (short)(U[(i / v) * 8 + (k / h)] - 128);
        const short vv = (short)(V[(i / v) * 8 + (k / h)] - 128);
        double dr = (double)yy + 1.280 * (double)vv;
        double dg = (double)yy - 0.215 * (double)uu;
        dg = dg - 0.381 * (double)vv;
        double db = (double)yy + 2.128 * (double)uu;
        int r = (int)dr, g = (int)dg, b = (int)db;
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        g = g < 0 ? 0 : (g > 255 ? 255 : g);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        out[t] = (u32)b | ((u32)g << 8) | ((u32)r << 16) | 0xff000000u;
    }
}

/* -------------------------------------------------------------- host side */

static void *g_stage = nullptr; /* 8 KiB device staging buffer */
#define STAGE_BYTES 8192

static int stage_ready(void)
{
    if (!ffhip_have_device()) return 0;
    if (!g_stage && hipMalloc(&g_stage, STAGE_BYTES) != hipSuccess) {
        g_stage = nullptr;
        return 0;
    }
    return 1;
}

static void fail(const char *what)
{
    /* the reference's op tables return void: nothing to report through; say it once */
    static int said = 0;
    if (!said || FFHIP_ENV("FFHIP_VERBOSE")) fprintf(stderr, "ffpic_hip: %s failed; block left untouched\n", what);
    said = 1;
}

static void run_inplace(int16_t *in, size_t bytes, int which)
{
    if (!stage_ready()) { fail("device init"); return; }
    if (hipMemcpy(g_stage, in, bytes, hipMemcpyHostToDevice) != hipSuccess) { fail("h2d"); return; }
    if (which == 8) hipLaunchKernelGGL(k_block_idct8x8, dim3(1), dim3(64), 0, 0, (short *)g_stage, 1);
    else hipLaunchKernelGGL(k_block_vp8_idct4x4, dim3(1), dim3(64), 0, 0, (short *)g_stage, 1);
    if (hipGetLastError() != hipSuccess) { fail("launch"); return; }
    if (hipMemcpy(in, g_stage, bytes, hipMemcpyDeviceToHost) != hipSuccess) fail("d2h");
}

static void hip_idct_4x4(int16_t *in, int bitdepth) { (void)bitdepth; run_inplace(in, 32, 4); }
static void hip_idct_8x8(int16_t *in, int bitdepth) { (void)bitdepth; run_inplace(in, 128, 8); }
static void hip_idct_4x4_v(void *in, int bitdepth) { hip_idct_4x4((int16_t *)in, bitdepth); }
static void hip_idct_8x8_v(void *in, int bitdepth) { hip_idct_8x8((int16_t *)in, bitdepth); }

extern "C" void ffhip_idct_4x4_hevc(const int16_t *in, int16_t *out, int bitdepth, bool epp)
{
    if (!stage_ready()) { fail("device init"); return; }
    short *d = (short *)g_stage;
    if (hipMemcpy(d, in, 32, hipMemcpyHostToDevice) != hipSuccess) { fail("h2d"); return; }
    hipLaunchKernelGGL(k_block_hevc_dst4x4, dim3(1), dim3(64), 0, 0, d, d + 16, 1, bitdepth, epp ? 1 : 0);
    if (hipGetLastError() != hipSuccess) { fail("launch"); return; }
    if (hipMemcpy(out, d + 16, 32, hipMemcpyDeviceToHost) != hipSuccess) fail("d2h");
}

static void hip_yuv_to_bgra32(uint8_t *dst, int pitch, void *Y, void *U, void *V, int v, int h)
{
    /* any pair the reference's caller can pass: its MCU scratch holds h*v <= 4 luma blocks (jpg.c:501); a pair beyond
     * that would make the reference itself read past Y[], so there is no behaviour to match */
    if (v < 1 || h < 1 || h * v > 4) { fail("sampling factor (h*v > 4)"); return; }
    if (!stage_ready()) { fail("device init"); return; }
    /* staging layout: Y (h*v*128 B) | U 128 | V 128 | out (8v*8h*4 B <= 1024) */
    char *d = (char *)g_stage;
    const size_t yb = (size_t)h * v * 128;
    if (hipMemcpy(d, Y, yb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + 512, U, 128, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + 640, V, 128, hipMemcpyHostToDevice) != hipSuccess) { fail("h2d"); return; }
    hipLaunchKernelGGL(k_mcu_color, dim3(1), dim3(256), 0, 0, (const short *)d, (const short *)(d + 512),
                       (const short *)(d + 640), v, h, (u32 *)(d + 1024));
    if (hipGetLastError() != hipSuccess) { fail("launch"); return; }
    uint8_t tmp[1024];
    if (hipMemcpy(tmp, d + 1024, (size_t)64 * v * h * 4, hipMemcpyDeviceToHost) != hipSuccess) { fail("d2h"); return; }
    for (int i = 0; i < 8 * v; i++) memcpy(dst + (size_t)i * pitch, tmp + (size_t)i * 8 * h * 4, (size_t)8 * h * 4);
}

static struct ffhip_accl_ops g_accl = {hip_idct_4x4, hip_idct_8x8, GPU_TYPE_HIP, {nullptr, nullptr}};
static int g_registered = 0;
static const struct ffhip_dct_ops g_dct16 = {16, hip_idct_4x4_v, hip_idct_8x8_v, nullptr, nullptr};
static const struct ffhip_cs_ops g_cs16 = {hip_yuv_to_bgra32, nullptr};

extern "C" struct ffhip_accl_ops *ffhip_accl_ops_get(void) { return stage_ready() ? &g_accl : nullptr; }

extern "C" void hip_accl_init(void)
{
    /* arch/opencl/opcl.c:112-114: a back-end that cannot run does not register */
    if (!stage_ready() || g_registered) return;
    typedef void (*reg_fn)(void *);
    reg_fn reg = (reg_fn)dlsym(RTLD_DEFAULT, "accl_ops_register"); /* arch/accl.c:17-19 */
    if (reg) {
        reg(&g_accl);
        g_registered = 1;
    }
}

extern "C" void hip_accl_uninit(void)
{
    /* like opcl_amd_uninit / vulkan_uninit (arch/accl.c:54-62): releases device
     * resources; the registry never unlinks entries, so the ops struct stays valid
     * (static storage) and re-acquires the staging buffer on its next call */
    if (g_stage) {
        (void)hipFree(g_stage);
        g_stage = nullptr;
    }
}

extern "C" const struct ffhip_dct_ops *ffhip_get_dct_ops(int component_bits)
{
    if ((component_bits - 1) / 8 != 1) return nullptr; /* idct.c:829-832 index 1 = 16-bit table */
    return stage_ready() ? &g_dct16 : nullptr;
}

extern "C" const struct ffhip_cs_ops *ffhip_get_cs_ops(int component_bits)
{
    if ((component_bits - 1) / 8 != 1) return nullptr;
    return stage_ready() ? &g_cs16 : nullptr;
}
