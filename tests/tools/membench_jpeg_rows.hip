// membench_jpeg_rows.hip -- round 6: does the SHAPE of the fused JPEG kernel's stores decide the placement lottery?
// The shipped kernel's wave stores its quad as 16 rows x 256 B (four instructions of 8 rows x 128 B); here the same bytes
// are stored, with no arithmetic, in other shapes, on SEVERAL allocations of the output buffer held at once:
//   E  8 rows x 128 B per store instruction (the shipped pattern), XCD-contiguous chunks
//   G  the four waves of a workgroup (4 adjacent quads = 256 px x 16 rows) exchange rows: wave w stores rows 4w .. 4w+3,
//      every store instruction ONE row x 1024 B contiguous (what a copy kernel's stores look like)
//   H  two rows x 512 B per store instruction (waves pair up)
//   L  a linear 16 B/lane copy of the same number of bytes (the placement-independent reference)
// Diagnostic only; build: hipcc -O3 --offload-arch=gfx950 -o membench_jpeg_rows.bin membench_jpeg_rows.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(256) void k(const char* y, const char* u, const char* v, char* out, int qpr, int rows, long pitch, int n_img) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nb = gridDim.x, per = (nb + 7) / 8;
  long bid = (long)(blockIdx.x % 8) * per + blockIdx.x / 8;
  if (bid >= nb) return;
  const int wpi = (qpr * rows + 3) / 4;   /* workgroups per image */
  const int img = (int)(bid / wpi); const int wgi = (int)(bid - (long)img * wpi);
  if (img >= n_img) return;
  if (PAT == 3) { /* linear copy of 7 B/px worth: 3 B/px read, 4 B/px written, 16 B per lane */
    const long px = (long)qpr * 64 * rows * 16; /* per image */
    const long wbytes = px * 4, per_wg = 16 * 256 * 4 * 4; /* a workgroup's output share: 4 quads x 4096 B... = 16 KB */
    const long o0 = (long)img * wbytes + (long)wgi * per_wg;
    const u32x4 a = __builtin_nontemporal_load((const u32x4*)(y + ((long)img * qpr * rows * 4 + (long)wgi * 16) * 512 + threadIdx.x * 16));
    const u32x4 b = __builtin_nontemporal_load((const u32x4*)(y + ((long)img * qpr * rows * 4 + (long)wgi * 16) * 512 + 4096 + threadIdx.x * 16));
    const u32x4 c = __builtin_nontemporal_load((const u32x4*)(u + ((long)img * qpr * rows * 4 + (long)wgi * 16) * 128 + (threadIdx.x & 127) * 16));
    const u32x4 s = a ^ b ^ c;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
      if (o0 + kk * 4096 + threadIdx.x * 16 + 16 <= (long)n_img * wbytes)
        __builtin_nontemporal_store(s + (unsigned)kk, (u32x4*)(out + o0 + kk * 4096 + threadIdx.x * 16));
    return;
  }
  int q = wgi * 4 + wave;
  const bool live = q < qpr * rows;
  if (!live) q = qpr * rows - 1;
  const int mrow = q / qpr, qcol = q - mrow * qpr;
  const long mcu = ((long)img * rows + mrow) * (qpr * 4) + qcol * 4;
  const u32x4 a = __builtin_nontemporal_load((const u32x4*)(y + mcu * 512 + lane * 16));
  const u32x4 b = __builtin_nontemporal_load((const u32x4*)(y + mcu * 512 + 1024 + lane * 16));
  const u32x4 c = __builtin_nontemporal_load((const u32x4*)((lane < 32 ? u : v) + mcu * 128 + (lane & 31) * 16));
  const u32x4 s = a ^ b ^ c;
  if (PAT == 0) {
    if (!live) return;
    char* o = out + (long)img * pitch * rows * 16 + (long)mrow * 16 * pitch + (long)qcol * 256;
#pragma unroll
    for (int rnd = 0; rnd < 2; rnd++)
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
        __builtin_nontemporal_store(s + (unsigned)(rnd + kk), (u32x4*)(o + (long)(kk * 8 + (lane >> 3)) * pitch + rnd * 128 + (lane & 7) * 16));
  } else {
    /* the workgroup's span: quads wgi*4 .. +3; when they lie in one MCU row (qpr % 4 == 0 here: 60) it is 1024 B x 16 rows */
    __shared__ u32x4 xch[4][4][64]; /* 16 KB: [wave][instruction][lane] -- the exchange a real kernel would need */
#pragma unroll
    for (int kk = 0; kk < 4; kk++) xch[wave][kk][lane] = s + (unsigned)kk;
    __syncthreads();
    const int q0 = wgi * 4, mrow0 = q0 / qpr, qcol0 = q0 - mrow0 * qpr;
    char* o = out + (long)img * pitch * rows * 16 + (long)mrow0 * 16 * pitch + (long)qcol0 * 256;
    if (PAT == 1) { /* wave w: rows 4w .. 4w+3, one row of 1024 B per instruction */
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const u32x4 t = xch[lane >> 4][kk][(lane & 15) * 4 + wave]; /* any permutation: the bytes' values do not matter here */
        __builtin_nontemporal_store(t, (u32x4*)(o + (long)(wave * 4 + kk) * pitch + lane * 16));
      }
    } else {        /* two rows of 512 B per instruction: waves 0,1 the left half, 2,3 the right half; rows (w & 1) * 8 + 2 kk + (lane >> 5) */
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const u32x4 t = xch[lane >> 4][kk][(lane & 15) * 4 + wave];
        __builtin_nontemporal_store(t, (u32x4*)(o + (long)((wave & 1) * 8 + 2 * kk + (lane >> 5)) * pitch + (wave >> 1) * 512 + (lane & 31) * 16));
      }
    }
  }
}
int main(int argc, char** argv) {
  const int cols = 240, rows = 135, n = 256, qpr = cols / 4;
  const int n_alloc = argc > 1 ? atoi(argv[1]) : 6;
  const long pad = argc > 2 ? atol(argv[2]) : 0;
  const long mcus = (long)cols * rows * n, pitch = cols * 64L + pad;
  char *y, *u, *v, *out[16];
  hipMalloc(&y, mcus * 512); hipMalloc(&u, mcus * 128); hipMalloc(&v, mcus * 128);
  hipMemset(y, 1, mcus * 512); hipMemset(u, 2, mcus * 128); hipMemset(v, 3, mcus * 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double px = (double)cols * 16 * rows * 16 * n;
  const dim3 grid((unsigned)(((qpr * rows + 3) / 4) * n), 1, 1);
#define RUN(P, O) ({ for (int w = 0; w < 2; w++) k<P><<<grid, 256>>>(y, u, v, O, qpr, rows, pitch, n); \
    hipEventRecord(e0); for (int r = 0; r < 6; r++) k<P><<<grid, 256>>>(y, u, v, O, qpr, rows, pitch, n); \
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 6; 7 * px / ms / 1e6; })
  for (int a = 0; a < n_alloc && a < 16; a++) {
    if (hipMalloc(&out[a], pitch * rows * 16 * n) != hipSuccess) { printf("alloc %d failed\n", a); break; }
    for (int rep = 0; rep < 2; rep++) {
      const double e = RUN(0, out[a]), g = RUN(1, out[a]), h = RUN(2, out[a]), l = RUN(3, out[a]);
      printf("alloc %2d %p pitch %ld: E 8x128B %.0f | G 1x1024B %.0f | H 2x512B %.0f | L linear %.0f GB/s\n", a, (void*)out[a], pitch, e, g, h, l);
      fflush(stdout);
    }
  }
  return 0;
}
