// membench.hip -- HBM streaming ceilings on this box: copy / read-only / write-only with
// 16 B per lane, several grid sizes, unroll factors and cache policies.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/membench.hip -o tests/tools/membench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(256) void k_copy(u32x4* __restrict__ d, const u32x4* __restrict__ s, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(&s[i + u * stride]) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v[u], &d[i + u * stride]); else d[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) d[i] = s[i];
}
template <int U, int NT>
__global__ __launch_bounds__(256) void k_read(u32x4* __restrict__ d, const u32x4* __restrict__ s, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= NT ? __builtin_nontemporal_load(&s[i + u * stride]) : s[i + u * stride];
  }
  if (acc[0] == 0x12345678 && acc[1] == 0x9abcdef0) d[0] = acc;
}
template <int NT>
__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ d, size_t n) {
  size_t stride = (size_t)gridDim.x * 256;
  u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, &d[i]); else d[i] = v; }
}
// contiguous-chunk variant: each block streams its own contiguous range
template <int NT>
__global__ __launch_bounds__(256) void k_copy_chunk(u32x4* __restrict__ d, const u32x4* __restrict__ s, size_t n) {
  size_t per = (n + gridDim.x - 1) / gridDim.x;
  size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
  for (size_t i = b + threadIdx.x; i < e; i += 256) { u32x4 v = NT ? __builtin_nontemporal_load(&s[i]) : s[i]; if (NT) __builtin_nontemporal_store(v, &d[i]); else d[i] = v; }
}

#define TIME(name, bytes, launch)                                              \
  do {                                                                         \
    for (int w = 0; w < 2; w++) { launch; }                                    \
    hipEventRecord(e0);                                                        \
    for (int r = 0; r < 5; r++) { launch; }                                    \
    hipEventRecord(e1); hipEventSynchronize(e1);                               \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                \
    printf("%-34s %8.1f GB/s\n", name, (double)(bytes) * 5 / (ms * 1e-3) / 1e9); \
  } while (0)

int main() {
  size_t bytes = (size_t)4 << 30, n = bytes / 16;
  u32x4 *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
  char name[128];
  for (int g : grids) {
    snprintf(name, sizeof name, "copy u1 grid %d", g); TIME(name, 2 * bytes, (k_copy<1, 0><<<g, 256>>>(b, a, n)));
    snprintf(name, sizeof name, "copy u4 grid %d", g); TIME(name, 2 * bytes, (k_copy<4, 0><<<g, 256>>>(b, a, n)));
    snprintf(name, sizeof name, "copy u4 nt grid %d", g); TIME(name, 2 * bytes, (k_copy<4, 1><<<g, 256>>>(b, a, n)));
  }
  TIME("copy u1 grid full", 2 * bytes, (k_copy<1, 0><<<(unsigned)(n / 256), 256>>>(b, a, n)));
  TIME("copy u1 nt grid full", 2 * bytes, (k_copy<1, 1><<<(unsigned)(n / 256), 256>>>(b, a, n)));
  TIME("copy chunk grid 2048", 2 * bytes, (k_copy_chunk<0><<<2048, 256>>>(b, a, n)));
  TIME("copy chunk nt grid 2048", 2 * bytes, (k_copy_chunk<1><<<2048, 256>>>(b, a, n)));
  TIME("copy chunk grid 8192", 2 * bytes, (k_copy_chunk<0><<<8192, 256>>>(b, a, n)));
  TIME("hipMemcpyDtoD", 2 * bytes, hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
  for (int g : {2048, 8192}) {
    snprintf(name, sizeof name, "read u4 grid %d", g); TIME(name, bytes, (k_read<4, 0><<<g, 256>>>(b, a, n)));
    snprintf(name, sizeof name, "read u4 nt grid %d", g); TIME(name, bytes, (k_read<4, 1><<<g, 256>>>(b, a, n)));
    snprintf(name, sizeof name, "write grid %d", g); TIME(name, bytes, (k_write<0><<<g, 256>>>(b, n)));
    snprintf(name, sizeof name, "write nt grid %d", g); TIME(name, bytes, (k_write<1><<<g, 256>>>(b, n)));
  }
  return 0;
}
