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
 * per macroblock (6.25 B/pixel).  32 lanes own one macroblock and move its bytes linearly, 16 per lane
 * and instruction; blocks are assembled inside lane pairs (k_vp8_residual); the Y2 lane hands the 16 DC values
 * to the luma lanes of the same wave through LDS (in-order within a wave, no barrier).
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

/* Four dwords per lane, picked inside every lane pair in ONE instruction each (v_cndmask_b32 with a DPP source operand):
 * lanes in `own` keep s1, the others take s0 of the pair's even (FROM_ODD = false: quad_perm [0,0,2,2]) or odd ([1,1,3,3])
 * lane.  A v_mov_b32_dpp broadcast followed by a v_cndmask, as the compiler writes the same thing, is twice the VALU
 * work: 32 of this kernel's 276 instructions.  (s_nop 1: the two wait states a DPP read needs after the VALU write of
 * its source, which the hazard pass cannot insert inside an asm block.) */
template <bool FROM_ODD>
__device__ __forceinline__ u32x4 pair_pick(u32x4 s0, u32x4 s1, unsigned long long own)
{
    u32 d0, d1, d2, d3;
    if (FROM_ODD)
        asm("s_mov_b64 vcc, %12\n\ts_nop 1\n\t"
            "v_cndmask_b32_dpp %0, %4, %8, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %1, %5, %9, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %2, %6, %10, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %3, %7, %11, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
            : "v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(s1[0]), "v"(s1[1]), "v"(s1[2]), "v"(s1[3]), "s"(own)
            : "vcc");
    else
        asm("s_mov_b64 vcc, %12\n\ts_nop 1\n\t"
            "v_cndmask_b32_dpp %0, %4, %8, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %1, %5, %9, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %2, %6, %10, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %3, %7, %11, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
            : "v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(s1[0]), "v"(s1[1]), "v"(s1[2]), "v"(s1[3]), "s"(own)
            : "vcc");
    return u32x4{d0, d1, d2, d3};
}

/* 32 lanes own one macroblock.  Memory moves LINEARLY: lane t loads 16-byte chunk t of the macroblock's 800 bytes of
 * levels and chunk 32 + t (t < 18), and stores chunk t and chunk 32 + t (t < 16) of its 768 bytes of residual, so
 * every vector memory instruction covers one contiguous run -- with a block (two chunks) per lane straight from
 * memory, each instruction touched half of every 64-byte segment, which held the kernel at 5.1 TB/s where the same
 * arithmetic over linear accesses measures 5.7-5.9.  Blocks are then assembled inside lane PAIRS with two DPP
 * broadcasts: even lane 2b computes luma block b from chunks 2b (its own) and 2b + 1 (its neighbour's); odd lane
 * 2j + 1 computes block 16 + j (U/V blocks 16-23, the Y2 block 24 on lane 17) from chunks 32 + 2j (its neighbour's
 * second load) and 32 + 2j + 1 (its own).  The results go back the same way (pair_pick). */
template <bool PATTERN> /* PATTERN: the loads and the stores without the arithmetic (diagnostics, FFHIP_VP8_RESIDUAL_PATTERN=1: the kernel's access pattern as its ceiling) */
__global__ __launch_bounds__(256) void k_vp8_residual(Vp8ResArgs a)
{
    __shared__ __attribute__((aligned(16))) short y2in[8][16];
    __shared__ __attribute__((aligned(16))) int y2t[8][16];
    const int t = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const long long mb = (long long)blockIdx.x * 8 + slot;
    if (mb >= a.n_mb) return; /* whole 32-lane slots: the lane pairs below are never split */
    const bool even = (t & 1) == 0;
    const int blk = even ? t >> 1 : 16 + (t >> 1); /* the block this lane computes; >= 25: none (odd lanes 19..31) */
    const bool works = blk < 25;
    const int kb = works ? blk : 0;
    /* the macroblock's 32 info bytes as two aligned dwords per lane (the one holding nz[blk], and bytes 24-27: nz of the
     * Y2 block, has_y2, segment) and the quantiser pair as one dword */
    const u32 *info = (const u32 *)(a.info + mb * 32);
    const u32 iw = info[kb >> 2], ic = info[6];
    const int nz = (int)((iw >> (8 * (kb & 3))) & 0xffu), nz24 = (int)(ic & 0xffu), has_y2 = ((ic >> 8) & 0xffu) != 0, seg = (int)((ic >> 16) & 3u);
    const int qsel = kb < 16 ? 0 : (kb < 24 ? 4 : 2);
    const u32 qpair = *(const u32 *)(a.quant + seg * 8 + qsel); /* (dc, ac) of this lane's block kind */
    const u32 qdc = qpair & 0xffffu, qac = qpair >> 16;
    const u32x4 *src = (const u32x4 *)(a.levels + mb * 400);
    const u32x4 c1 = __builtin_nontemporal_load(src + t);
    u32x4 c2 = {0u, 0u, 0u, 0u};
    if (t < 18) c2 = __builtin_nontemporal_load(src + 32 + t);
    if (PATTERN) { /* (info and quantiser words are loaded as ever: iw, ic, qpair feed the stored words so that nothing is dropped) */
        u32x4 *dstp = (u32x4 *)(a.out + mb * 384);
        const u32 k = iw ^ ic ^ qpair;
        __builtin_nontemporal_store(c1 + k, dstp + t);
        if (t < 16) __builtin_nontemporal_store(c2 + k, dstp + 32 + t);
        return;
    }
    /* block assembly inside the lane pair */
    const unsigned long long even_lanes = 0x5555555555555555ull;
    const u32x4 l0 = pair_pick<false>(c2, c1, even_lanes);  /* even: my first chunk; odd: my even neighbour's second load */
    const u32x4 l1 = pair_pick<true>(c1, c2, ~even_lanes);  /* even: my odd neighbour's first chunk; odd: my second load */
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
     * of this VALU-bound kernel).  Instead the Y2 lane parks its 16 dequantised coefficients in LDS and the lane of luma
     * block 4r + i computes t[4r + i] of the column pass from column i, parks that, and computes w[4r + i] of the
     * row pass from row r: each is one of four +- combinations picked by r (then i), and w[blk] is exactly the DC
     * that block needs.  Writers and readers never sit on two sides of one branch (divergent sides have no defined
     * order): stores are predicated blocks followed by a wave-level fence; LDS serves a wave in program order. */
    if (blk == 24 && has_y2) {
        *(u32x4 *)&y2in[slot][0] = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *(u32x4 *)&y2in[slot][8] = u32x4{pk[4], pk[5], pk[6], pk[7]};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool luma = even; /* blk < 16 */
    const int r4 = (blk >> 2) & 3, i4 = blk & 3;
    int tv = 0;
    if (luma && has_y2) {
        const int v0 = y2in[slot][i4], v1 = y2in[slot][4 + i4], v2 = y2in[slot][8 + i4], v3 = y2in[slot][12 + i4];
        const int a4 = v0 + v3, b4 = v1 + v2, e4 = v1 - v2, f4 = v0 - v3;
        const int p4 = (r4 & 1) ? f4 : a4, q4 = (r4 & 1) ? e4 : b4;
        tv = (r4 & 2) ? p4 - q4 : p4 + q4; /* rows: a+b, f+e, a-b, f-e */
        y2t[slot][blk] = tv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (luma && has_y2) {
        const u32x4 row = *(const u32x4 *)&y2t[slot][4 * r4];
        const int t0 = (int)row[0], t1 = (int)row[1], t2 = (int)row[2], t3 = (int)row[3];
        const int a4 = t0 + t3, b4 = t1 + t2, e4 = t1 - t2, f4 = t0 - t3;
        const int p4 = (i4 & 1) ? f4 : a4, q4 = (i4 & 1) ? e4 : b4;
        const int full = (short)((((i4 & 2) ? p4 - q4 : p4 + q4) + 3) >> 3);
        const int fast = (short)((y2in[slot][0] + 3) >> 3); /* IWHT_fast, webp.c:1098-1106 */
        pk[0] = __builtin_amdgcn_perm(pk[0], (u32)(nz24 > 1 ? full : fast), 0x07060100u);
    }
    if (works && blk != 24 && (nz > 1 || (pk[0] & 0xffffu) != 0)) vp8_idct4x4(pk);
    /* back to chunks: chunk t = even ? my lower half : my even neighbour's upper half; chunk 32 + t (t < 16) = even ? my odd
     * neighbour's lower half : my upper half */
    const u32x4 o0 = {pk[0], pk[1], pk[2], pk[3]}, o1 = {pk[4], pk[5], pk[6], pk[7]};
    const u32x4 s1 = pair_pick<false>(o1, o0, even_lanes), s2 = pair_pick<true>(o0, o1, ~even_lanes);
    u32x4 *dst = (u32x4 *)(a.out + mb * 384);
    __builtin_nontemporal_store(s1, dst + t);
    if (t < 16) __builtin_nontemporal_store(s2, dst + 32 + t);
}

extern "C" int ffhip_vp8_residual_batch(long long n_mb, const int16_t *d_levels, const uint8_t *d_mbinfo,
                                        const uint16_t *d_quant, int16_t *d_residual, void *stream)
{
    if (n_mb < 0) return FFHIP_EINVAL;
    if (n_mb == 0) return FFHIP_OK;
    if (!d_levels || !d_mbinfo || !d_quant || !d_residual) return FFHIP_EINVAL;
    if (((uintptr_t)d_levels & 15) || ((uintptr_t)d_residual & 15) || ((uintptr_t)d_mbinfo & 3) || ((uintptr_t)d_quant & 3) || n_mb > 0x7fffffffLL * 8)
        return FFHIP_EINVAL;
    if (!ffhip_have_device()) return FFHIP_ENODEV;
    Vp8ResArgs a = {d_levels, d_mbinfo, d_quant, d_residual, n_mb};
    if (FFHIP_ENV("FFHIP_VP8_RESIDUAL_PATTERN")) hipLaunchKernelGGL(k_vp8_residual<true>, dim3((unsigned)((n_mb + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_vp8_residual<false>, dim3((unsigned)((n_mb + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
    FFHIP_CHECK(hipGetLastError(), FFHIP_EIO);
    return FFHIP_OK;
}
