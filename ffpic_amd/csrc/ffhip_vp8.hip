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

/* x is an int16 value here and k < 2^16: the product fits 32 bits and is a full-rate v_mul_i32_i24 (v_mul_lo_u32 runs at a quarter of the rate) */
__device__ __forceinline__ int vp8_mul(int x, int k) { return __mul24(x, k) >> 16; }

__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)); }
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)); }
/* ((x * k) >> 16) of both int16 halves: two 24-bit multiplies, the shift is the byte pick of one v_perm */
__device__ __forceinline__ u32 vp8_mul_pair(u32 x, int k)
{
    const int plo = __mul24((int)(short)(x & 0xffffu), k), phi = __mul24((int)x >> 16, k);
    return __builtin_amdgcn_perm((u32)phi, (u32)plo, 0x07060302u);
}

/* utils/idct.c:100-151 on packed pairs p[2r + h] = (c[4r + 2h], c[4r + 2h + 1]).  The vertical pass only ever
 * keeps int16 (idct.c:124 truncates its results), so it runs on two columns at a time in 16-bit lanes: sums
 * wrap exactly like the low halves of the reference's int sums, and every (x * k) >> 16 fits 16 bits.  The
 * horizontal pass needs the bits above 16 for its (.. + 4) >> 3 and stays in 32 bits. */
__device__ __forceinline__ void vp8_idct4x4(u32 p[8])
{
    u32 T[8];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const u32 x0 = p[h], x1 = p[2 + h], x2 = p[4 + h], x3 = p[6 + h];
        const u32 s = pk_add(x0, x2), d = pk_sub(x0, x2);
        const u32 lo = pk_sub(pk_sub(vp8_mul_pair(x1, 35468), x3), vp8_mul_pair(x3, 20091));
        const u32 hi = pk_add(pk_add(x1, vp8_mul_pair(x1, 20091)), vp8_mul_pair(x3, 35468));
        T[h] = pk_add(s, hi); T[2 + h] = pk_add(d, lo); T[4 + h] = pk_sub(d, lo); T[6 + h] = pk_sub(s, hi);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int x0 = (short)(T[2 * r] & 0xffffu), x1 = (int)T[2 * r] >> 16, x2 = (short)(T[2 * r + 1] & 0xffffu), x3 = (int)T[2 * r + 1] >> 16;
        const int s = x0 + x2, d = x0 - x2;
        const int lo = vp8_mul(x1, 35468) - x3 - vp8_mul(x3, 20091);
        const int hi = x1 + vp8_mul(x1, 20091) + vp8_mul(x3, 35468);
        p[2 * r] = __builtin_amdgcn_perm((u32)((d + lo + 4) >> 3), (u32)((s + hi + 4) >> 3), 0x05040100u);
        p[2 * r + 1] = __builtin_amdgcn_perm((u32)((s - hi + 4) >> 3), (u32)((d - lo + 4) >> 3), 0x05040100u);
    }
}

__global__ __launch_bounds__(256) void k_vp8_residual(Vp8ResArgs a)
{
    __shared__ __attribute__((aligned(16))) short y2in[8][16];
    __shared__ __attribute__((aligned(16))) int y2t[8][16];
    const int t = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const long long mb = (long long)blockIdx.x * 8 + slot;
    if (mb >= a.n_mb || t >= 25) return;
    const uint8_t *info = a.info + mb * 32;
    const int nz = info[t], nz24 = info[24], has_y2 = info[25] != 0, seg = info[26] & 3;
    const int qsel = t < 16 ? 0 : (t < 24 ? 4 : 2);
    const u32 qdc = a.quant[seg * 8 + qsel], qac = a.quant[seg * 8 + qsel + 1];
    const u32x4 *src = (const u32x4 *)(a.levels + (mb * 25 + t) * 16);
    const u32x4 l0 = __builtin_nontemporal_load(src), l1 = __builtin_nontemporal_load(src + 1);
    const u32 lv[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    u32 pk[8]; /* pk[2r + h] = (c[4r + 2h], c[4r + 2h + 1]) */
#pragma unroll
    for (int i = 0; i < 8; i++) {
        /* low 16 bits of level*q == the int16 store of webp.c:1061 */
        const u32 f = i == 0 ? (qdc | (qac << 16)) : (qac | (qac << 16));
        using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
        const u32 lvi = lv[i];
        pk[i] = __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, lvi) * __builtin_bit_cast(u16x2, f)));
    }
    /* Y2 -> luma DCs, ACROSS the sixteen luma lanes of the macroblock (webp.c:1067-1106).  Done by the Y2 lane
     * alone the inverse WHT is ~100 instructions that the whole wave pays for one or two working lanes (a third
     * of this VALU-bound kernel).  Instead lane 24 parks its 16 dequantised coefficients in LDS and luma lane
     * t = 4r + i computes t[4r + i] of the column pass from column i, parks that, and computes w[4r + i] of the
     * row pass from row r: each is one of four +- combinations picked by r (then i), and w[t] is exactly the DC
     * lane t needs.  Writers and readers never sit on two sides of one branch (divergent sides have no defined
     * order): stores are predicated blocks followed by a wave-level fence; LDS serves a wave in program order. */
    if (t == 24 && has_y2) {
        *(u32x4 *)&y2in[slot][0] = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *(u32x4 *)&y2in[slot][8] = u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t == 24) return;
    const int r4 = (t >> 2) & 3, i4 = t & 3;
    int tv = 0;
    if (t < 16 && has_y2) {
        const int v0 = y2in[slot][i4], v1 = y2in[slot][4 + i4], v2 = y2in[slot][8 + i4], v3 = y2in[slot][12 + i4];
        const int a4 = v0 + v3, b4 = v1 + v2, e4 = v1 - v2, f4 = v0 - v3;
        const int p4 = (r4 & 1) ? f4 : a4, q4 = (r4 & 1) ? e4 : b4;
        tv = (r4 & 2) ? p4 - q4 : p4 + q4; /* rows: a+b, f+e, a-b, f-e */
        y2t[slot][t] = tv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t < 16 && has_y2) {
        const u32x4 row = *(const u32x4 *)&y2t[slot][4 * r4];
        const int t0 = (int)row[0], t1 = (int)row[1], t2 = (int)row[2], t3 = (int)row[3];
        const int a4 = t0 + t3, b4 = t1 + t2, e4 = t1 - t2, f4 = t0 - t3;
        const int p4 = (i4 & 1) ? f4 : a4, q4 = (i4 & 1) ? e4 : b4;
        const int full = (short)((((i4 & 2) ? p4 - q4 : p4 + q4) + 3) >> 3);
        const int fast = (short)((y2in[slot][0] + 3) >> 3); /* IWHT_fast, webp.c:1098-1106 */
        pk[0] = __builtin_amdgcn_perm(pk[0], (u32)(nz24 > 1 ? full : fast), 0x07060100u);
    }
    if (nz > 1 || (pk[0] & 0xffffu) != 0) vp8_idct4x4(pk);
    const u32x4 o0 = {pk[0], pk[1], pk[2], pk[3]}, o1 = {pk[4], pk[5], pk[6], pk[7]};
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
