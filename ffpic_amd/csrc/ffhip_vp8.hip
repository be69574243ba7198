/*
 * ffhip_vp8.hip -- VP8 (WebP lossy) residual stage for batches of macroblocks:
 * dequantisation, inverse WHT of the Y2 block and the 4x4 inverse DCTs, bit-exact with
 *   vp8_get_coefficients' dequant store   format/webp.c:1061
 *   IWHT_long / IWHT_fast                 format/webp.c:1067-1106
 *   idct_4x4_16                           utils/idct.c:100-151
 *   the per-MB assembly                   format/webp.c:1147-1196
 * including the reference's rule that a block is transformed only when more than one
 * token was read or its DC is non-zero (webp.c:1172,1188).
 *
 * HBM-bound byte/integer work: 800 B of levels + 32 B of info in, 768 B of residual out
 * per macroblock (6.25 B/pixel).  32 lanes own one macroblock: lane t < 16 = luma block t,
 * 16..23 = U/V blocks, 24 = the Y2 block; the Y2 lane hands the 16 DC values to the luma
 * lanes of the same wave through LDS (in-order within a wave, no barrier).
 */
#include "ffhip_internal.h"

struct Vp8ResArgs {
    const int16_t *levels; /* [n_mb][25][16] */
    const uint8_t *info;   /* [n_mb][32]: nz[25], has_y2, segment */
    const uint16_t *quant; /* [4][8]: y1_dc y1_ac y2_dc y2_ac uv_dc uv_ac - - */
    int16_t *out;          /* [n_mb][24][16] */
    long long n_mb;
};

__device__ __forceinline__ int vp8_mul(int x, int k) { return (x * k) >> 16; }

/* utils/idct.c:100-151; pass-1 results are truncated to int16 (idct.c:124) */
__device__ __forceinline__ void vp8_idct4x4(int c[16])
{
    int t[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x0 = c[i], x1 = c[4 + i], x2 = c[8 + i], x3 = c[12 + i];
        const int s = x0 + x2, d = x0 - x2;
        const int lo = vp8_mul(x1, 35468) - x3 - vp8_mul(x3, 20091);
        const int hi = x1 + vp8_mul(x1, 20091) + vp8_mul(x3, 35468);
        t[i] = (short)(s + hi); t[4 + i] = (short)(d + lo); t[8 + i] = (short)(d - lo); t[12 + i] = (short)(s - hi);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int x0 = t[4 * r], x1 = t[4 * r + 1], x2 = t[4 * r + 2], x3 = t[4 * r + 3];
        const int s = x0 + x2, d = x0 - x2;
        const int lo = vp8_mul(x1, 35468) - x3 - vp8_mul(x3, 20091);
        const int hi = x1 + vp8_mul(x1, 20091) + vp8_mul(x3, 35468);
        c[4 * r] = (short)((s + hi + 4) >> 3); c[4 * r + 1] = (short)((d + lo + 4) >> 3);
        c[4 * r + 2] = (short)((d - lo + 4) >> 3); c[4 * r + 3] = (short)((s - hi + 4) >> 3);
    }
}

/* format/webp.c:1067-1096; w[k] is the DC of luma block k */
__device__ __forceinline__ void vp8_iwht(const int c[16], int w[16])
{
    int t[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int a = c[i] + c[12 + i], b = c[4 + i] + c[8 + i], e = c[4 + i] - c[8 + i], f = c[i] - c[12 + i];
        t[i] = a + b; t[4 + i] = f + e; t[8 + i] = a - b; t[12 + i] = f - e;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int a = t[4 * r] + t[4 * r + 3], b = t[4 * r + 1] + t[4 * r + 2];
        const int e = t[4 * r + 1] - t[4 * r + 2], f = t[4 * r] - t[4 * r + 3];
        w[4 * r] = (short)((a + b + 3) >> 3); w[4 * r + 1] = (short)((f + e + 3) >> 3);
        w[4 * r + 2] = (short)((a - b + 3) >> 3); w[4 * r + 3] = (short)((f - e + 3) >> 3);
    }
}

__global__ __launch_bounds__(256) void k_vp8_residual(Vp8ResArgs a)
{
    __shared__ short y2dc[8][16];
    const int t = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const long long mb = (long long)blockIdx.x * 8 + slot;
    if (mb >= a.n_mb || t >= 25) return;
    const uint8_t *info = a.info + mb * 32;
    const int nz = info[t], has_y2 = info[25] != 0, seg = info[26] & 3;
    const int qsel = t < 16 ? 0 : (t < 24 ? 4 : 2);
    const u32 qdc = a.quant[seg * 8 + qsel], qac = a.quant[seg * 8 + qsel + 1];
    const u32x4 *src = (const u32x4 *)(a.levels + (mb * 25 + t) * 16);
    const u32x4 l0 = __builtin_nontemporal_load(src), l1 = __builtin_nontemporal_load(src + 1);
    const u32 lv[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    int c[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        /* low 16 bits of level*q == the int16 store of webp.c:1061 */
        const u32 f = i == 0 ? (qdc | (qac << 16)) : (qac | (qac << 16));
        using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
        const u32 lvi = lv[i];
        const u32 p = __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, lvi) * __builtin_bit_cast(u16x2, f)));
        c[2 * i] = (int)(short)(p & 0xffffu);
        c[2 * i + 1] = (int)p >> 16;
    }
    /* Y2 -> luma DC hand-off inside the wave.  The writer (lane 24) and the readers (lanes
     * 0-15) must not sit on the two sides of one branch -- divergent sides have no defined
     * order -- so the store is a reconverging predicated block, followed by a wave-level
     * fence; LDS then serves the wave's accesses in program order. */
    if (t == 24 && has_y2) {
        int w[16];
        if (nz > 1) vp8_iwht(c, w);
        else {
            const int dc0 = (short)((c[0] + 3) >> 3); /* IWHT_fast, webp.c:1098-1106 */
#pragma unroll
            for (int k = 0; k < 16; k++) w[k] = dc0;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) y2dc[slot][k] = (short)w[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t == 24) return;
    if (t < 16 && has_y2) c[0] = y2dc[slot][t];
    if (nz > 1 || c[0] != 0) vp8_idct4x4(c);
    u32x4 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o0[i] = ((u32)c[2 * i] & 0xffffu) | ((u32)c[2 * i + 1] << 16);
        o1[i] = ((u32)c[8 + 2 * i] & 0xffffu) | ((u32)c[8 + 2 * i + 1] << 16);
    }
    u32x4 *dst = (u32x4 *)(a.out + (mb * 24 + t) * 16);
    __builtin_nontemporal_store(o0, dst);
    __builtin_nontemporal_store(o1, dst + 1);
}

extern "C" int ffhip_vp8_residual_batch(long long n_mb, const int16_t *d_levels, const uint8_t *d_mbinfo,
                                        const uint16_t *d_quant, int16_t *d_residual, void *stream)
{
    if (n_mb < 0) return FFHIP_EINVAL;
    if (n_mb == 0) return FFHIP_OK;
    if (!d_levels || !d_mbinfo || !d_quant || !d_residual) return FFHIP_EINVAL;
    if (((uintptr_t)d_levels & 15) || ((uintptr_t)d_residual & 15) || n_mb > 0x7fffffffLL * 8) return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    Vp8ResArgs a = {d_levels, d_mbinfo, d_quant, d_residual, n_mb};
    hipLaunchKernelGGL(k_vp8_residual, dim3((unsigned)((n_mb + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
