// probe_gfx950.hip -- prints what a few gfx950 primitives the kernels rely on really do.
// Diagnostic only (not part of the product or the test suite):
//   hipcc --offload-arch=gfx950 -O2 tests/tools/probe_gfx950.hip -o tests/tools/probe_gfx950.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;

__global__ void k_tr(const short* in, short* out) {
  __shared__ short lds[1024];
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  // plain [R][C] tile, 16 columns wide rows (32 B): lane 4q+p -> row q, cols 4p..4p+3
  int t = l & 15, q = t >> 2, p = t & 3, g = l >> 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + g * 64 + q * 16 + p * 4));
  for (int i = 0; i < 4; i++) out[l * 4 + i] = v[i];
}
__global__ void k_alu(int* out) {
  int l = threadIdx.x;
  if (l == 0) {
    s16x2 a = {32767, 32767}, b = {32767, 32767};
    out[0] = __builtin_amdgcn_sdot2(a, b, 0x7fffffff, false);     // wraps?
    s16x2 c = {-32768, -32768};
    out[1] = __builtin_amdgcn_sdot2(c, c, 5, false);
    u32 d; u32 s = (u32)(uint16_t)(-5) | ((u32)300 << 16);
    asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(s));
    out[2] = (int)d;                                              // expect 0x0000ff00
    u32 s2 = (u32)200 | ((u32)(uint16_t)(-32768) << 16);
    asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(s2));
    out[3] = (int)d;                                              // expect 0x000000c8
    out[4] = (int)__builtin_amdgcn_perm(0x44332211u, 0xddccbbaau, 0x05010400u); // S1 bytes aa bb cc dd = 0..3
    out[5] = (int)__builtin_amdgcn_perm(0x44332211u, 0xddccbbaau, 0x0d040100u);
    out[6] = (int)__builtin_amdgcn_perm(0x44332211u, 0xddccbbaau, 0x0d050302u);
  }
}
int main() {
  short h[1024], o[256]; for (int i = 0; i < 1024; i++) h[i] = (short)i;
  short *din, *dout; int* dalu; int alu[8];
  hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o); hipMalloc(&dalu, sizeof alu);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, din, dout);
  hipLaunchKernelGGL(k_alu, dim3(1), dim3(64), 0, 0, dalu);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost); hipMemcpy(alu, dalu, sizeof alu, hipMemcpyDeviceToHost);
  printf("tr16 (tile [4 rows][16 cols] of element index r*16+c, group 0): expect lane i = {i, 16+i, 32+i, 48+i}\n");
  for (int l = 0; l < 20; l++) printf(" lane %2d: %4d %4d %4d %4d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  printf("sdot2 wrap: %d (expect %d) ; %d (expect %d)\n", alu[0], (int)(uint32_t)(2u*32767u*32767u + 0x7fffffffu), alu[1], (int)(2147483648u + 5u));
  printf("sat_pk: %08x (expect 0000ff00) %08x (expect 000000c8)\n", alu[2], alu[3]);
  printf("perm: %08x (expect 22bb11aa) %08x (expect ff11bbaa) %08x (expect ff22ddcc)\n", alu[4], alu[5], alu[6]);
  return 0;
}
