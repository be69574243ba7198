// valu_rate.hip -- issue rate of the VALU instructions the fused JPEG kernel is made of, with
// 8 waves per SIMD resident (cycles per wave-instruction per SIMD).  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/valu_rate.hip -o tests/tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

#define KERNEL(name, asmline)                                                          \
  __global__ __launch_bounds__(512) void name(u32 *out, int iters) {                   \
    u32 a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 9, a5 = a0 + 11, a6 = a0 ^ 5, a7 = a0 ^ 77; \
    u32 b = threadIdx.x * 2654435761u + 12345u;                                         \
    for (int i = 0; i < iters; i++) {                                                  \
      REP16(asm volatile(asmline : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) \
    }                                                                                  \
    out[blockIdx.x * 512 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;      \
  }
// each asm line = 8 independent instructions
#define L8(op) op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n"
#define L8_3(op) op " %0, %8, %0, %0\n" op " %1, %8, %1, %1\n" op " %2, %8, %2, %2\n" op " %3, %8, %3, %3\n" op " %4, %8, %4, %4\n" op " %5, %8, %5, %5\n" op " %6, %8, %6, %6\n" op " %7, %8, %7, %7\n"
#define L8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
KERNEL(k_add, L8("v_add_u32"))
KERNEL(k_dot2c, L8("v_dot2c_i32_i16"))
KERNEL(k_pkmul, L8("v_pk_mul_lo_u16"))
KERNEL(k_pkadd, L8("v_pk_add_u16"))
KERNEL(k_pkmax, L8("v_pk_max_i16"))
KERNEL(k_pkashr, L8("v_pk_ashrrev_i16"))
KERNEL(k_perm, L8_3("v_perm_b32"))
KERNEL(k_satpk, L8_1("v_sat_pk_u8_i16"))
KERNEL(k_lshl, L8("v_lshlrev_b32"))
KERNEL(k_mad24, L8_3("v_mad_i32_i24"))
KERNEL(k_mullo, L8("v_mul_lo_u32"))
KERNEL(k_mulf, L8("v_mul_f32"))
KERNEL(k_cvtfi, L8_1("v_cvt_f32_i32"))
KERNEL(k_cvtif, L8_1("v_cvt_i32_f32"))
KERNEL(k_lshlor, L8_3("v_lshl_or_b32"))
KERNEL(k_and, L8("v_and_b32"))
KERNEL(k_or, L8("v_or_b32"))
KERNEL(k_sub, L8("v_sub_u32"))
KERNEL(k_max, L8("v_max_i32"))
KERNEL(k_ashr, L8("v_ashrrev_i32"))
KERNEL(k_add3, L8_3("v_add3_u32"))
KERNEL(k_xor, L8("v_xor_b32"))
KERNEL(k_bfe, L8_3("v_bfe_i32"))
KERNEL(k_fma, L8_3("v_fma_f32"))
KERNEL(k_mov, L8_1("v_mov_b32"))
KERNEL(k_addf, L8("v_add_f32"))
KERNEL(k_mul24, L8("v_mul_i32_i24"))
KERNEL(k_andor, L8_3("v_and_or_b32"))
KERNEL(k_med3, L8_3("v_med3_i32"))

template <typename K> void run(const char *name, K k, int n_per_line) {
  u32 *out; hipMalloc(&out, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256 * 4; // 4 x 512-thread blocks per CU = 8 waves/SIMD
  k<<<blocks, 512>>>(out, 10);
  hipEventRecord(e0); k<<<blocks, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts_per_simd = (double)iters * 16 * n_per_line * (blocks * 8.0) / 1024.0; // wave-instructions per SIMD
  printf("%-18s %8.3f ms  -> %.2f ns per wave-instr per SIMD (x clock GHz = cycles)\n", name, ms, ms * 1e6 / insts_per_simd);
  hipFree(out);
}
int main() {
  run("v_add_u32", k_add, 8); run("v_dot2c_i32_i16", k_dot2c, 8); run("v_pk_mul_lo_u16", k_pkmul, 8);
  run("v_pk_add_u16", k_pkadd, 8); run("v_pk_max_i16", k_pkmax, 8); run("v_pk_ashrrev_i16", k_pkashr, 8);
  run("v_perm_b32", k_perm, 8); run("v_sat_pk_u8_i16", k_satpk, 8); run("v_lshlrev_b32", k_lshl, 8);
  run("v_mad_i32_i24", k_mad24, 8); run("v_mul_lo_u32", k_mullo, 8); run("v_mul_f32", k_mulf, 8);
  run("v_cvt_f32_i32", k_cvtfi, 8); run("v_cvt_i32_f32", k_cvtif, 8); run("v_lshl_or_b32", k_lshlor, 8);
  run("v_and_b32", k_and, 8); run("v_or_b32", k_or, 8); run("v_sub_u32", k_sub, 8); run("v_max_i32", k_max, 8);
  run("v_ashrrev_i32", k_ashr, 8); run("v_add3_u32", k_add3, 8); run("v_xor_b32", k_xor, 8); run("v_bfe_i32", k_bfe, 8);
  run("v_fma_f32", k_fma, 8); run("v_mov_b32", k_mov, 8); run("v_add_f32", k_addf, 8); run("v_mul_i32_i24", k_mul24, 8);
  run("v_and_or_b32", k_andor, 8); run("v_med3_i32", k_med3, 8);
  return 0;
}
