// membench_jpeg.hip -- the fused JPEG kernel's exact memory access pattern with no arithmetic:
// per wave 3 non-temporal 16 B/lane loads (2 KiB luma + 2 x 512 B chroma of one quad) and 4
// non-temporal 16 B/lane stores (8 rows x 128 B each, row pitch 15360 B).  Gives the memory-side
// ceiling for this pattern (7 B/pixel, 3:4 read:write).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(256) void k(const char* y, const char* u, const char* v, char* out, int qpr, int rows, long pitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bid = blockIdx.x;
  if (PAT == 4 || PAT == 5) { /* XCD-aware: blocks b, b+8, ... share an XCD; give each XCD a contiguous chunk */
    const int nb = gridDim.x, per = (nb + 7) / 8;
    bid = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (bid >= nb) return;   /* (only exact when nb % 8 == 0; good enough for the experiment) */
  }
  int q = bid * 4 + wave; const int img = blockIdx.z;
  if (q >= qpr * rows) return;
  int mrow = q / qpr, qcol = q - mrow * qpr;
  if (PAT == 2) { /* the 4 waves of a workgroup take the same quad column of 4 consecutive MCU rows */
    const int g = blockIdx.x; const int rg = g / qpr, qc = g - rg * qpr; mrow = rg * 4 + wave; qcol = qc; if (mrow >= rows) return;
  }
  const long mcu = ((long)img * rows + mrow) * (qpr * 4) + qcol * 4;
  const u32x4 a = __builtin_nontemporal_load((const u32x4*)(y + mcu * 512 + lane * 16));
  const u32x4 b = __builtin_nontemporal_load((const u32x4*)(y + mcu * 512 + 1024 + lane * 16));
  const u32x4 c = __builtin_nontemporal_load((const u32x4*)((lane < 32 ? u : v) + mcu * 128 + (lane & 31) * 16));
  char* o = out + (long)img * pitch * rows * 16 + (long)mrow * 16 * pitch + (long)qcol * 256;
  const u32x4 s = a ^ b ^ c;
  if (PAT == 1 || PAT == 5) { /* 4 rows x 256 B per store instruction */
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
      __builtin_nontemporal_store(s + (unsigned)kk, (u32x4*)(o + (long)(kk * 4 + (lane >> 4)) * pitch + (lane & 15) * 16));
  } else if (PAT == 3) { /* plain (temporal) stores, 8 rows x 128 B */
#pragma unroll
    for (int rnd = 0; rnd < 2; rnd++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
        *(u32x4*)(o + (long)(kk * 8 + (lane >> 3)) * pitch + rnd * 128 + (lane & 7) * 16) = s + (unsigned)(rnd + kk);
  } else {
#pragma unroll
    for (int rnd = 0; rnd < 2; rnd++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
        __builtin_nontemporal_store(s + (unsigned)(rnd + kk), (u32x4*)(o + (long)(kk * 8 + (lane >> 3)) * pitch + rnd * 128 + (lane & 7) * 16));
  }
}
int main() {
  const int cols = 240, rows = 135, n = 256, qpr = cols / 4;
  const long mcus = (long)cols * rows * n, pitch = cols * 64L;
  char *y, *u, *v, *out;
  hipMalloc(&y, mcus * 512); hipMalloc(&u, mcus * 128); hipMalloc(&v, mcus * 128); hipMalloc(&out, pitch * rows * 16 * n);
  hipMemset(y, 1, mcus * 512); hipMemset(u, 2, mcus * 128); hipMemset(v, 3, mcus * 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double px = (double)cols * 16 * rows * 16 * n;
#define RUN(P, G, NAME) do { dim3 grid G; for (int w = 0; w < 3; w++) k<P><<<grid, 256>>>(y, u, v, out, qpr, rows, pitch); \
    hipEventRecord(e0); for (int r = 0; r < 10; r++) k<P><<<grid, 256>>>(y, u, v, out, qpr, rows, pitch); \
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; \
    printf("%-44s %.4f ms -> %.1f GB/s (7 B/px)\n", NAME, ms, 7 * px / ms / 1e6); } while (0)
  for (int rep = 0; rep < 2; rep++) {
    RUN(0, ((qpr * rows + 3) / 4, 1, n), "A: 8 rows x 128 B nt stores (current)");
    RUN(1, ((qpr * rows + 3) / 4, 1, n), "B: 4 rows x 256 B nt stores");
    RUN(2, (qpr * ((rows + 3) / 4), 1, n), "C: WG = 4 MCU rows of one quad column");
    RUN(3, ((qpr * rows + 3) / 4, 1, n), "D: plain stores");
    RUN(4, ((qpr * rows + 3) / 4, 1, n), "E: XCD-contiguous chunks, nt stores");
    RUN(5, ((qpr * rows + 3) / 4, 1, n), "F: XCD-contiguous + 256 B rows");
  }
  return 0;
}
