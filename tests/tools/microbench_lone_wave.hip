// What one wavefront alone on a SIMD pays per DEPENDENT instruction on gfx950 (the cost model of the dependency-bound
// kernels: HEVC intra, VP8 prediction / loop filter).  Build: hipcc --offload-arch=gfx950 -O3 -o lone_wave.bin microbench_lone_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k(unsigned long long *out, int seed)
{
    __shared__ int lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (i * 7 + 3) & 1023;
    __syncthreads();
    unsigned long long c0, c1, r0, r1;
    int v = seed + lane;
    // (a) dependent VALU chain
    r0 = wall_clock64(); c0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(lane));
    c1 = clock64(); r1 = wall_clock64();
    if (lane == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    // (b) dependent SALU chain
    int s = seed;
    r0 = wall_clock64(); c0 = clock64();
#pragma unroll 64
    for (int i = 0; i < N; i++) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(s)); /* no SCC: the loop counter lives there */
    c1 = clock64(); r1 = wall_clock64();
    if (lane == 0) { out[2] = c1 - c0; out[3] = r1 - r0; }
    // (c) dependent LDS read chain
    int p = lane;
    r0 = wall_clock64(); c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N / 4; i++) p = lds[p];
    c1 = clock64(); r1 = wall_clock64();
    if (lane == 0) { out[4] = c1 - c0; out[5] = r1 - r0; }
    // (d) alternating SALU / VALU dependent (readfirstlane round trips)
    r0 = wall_clock64(); c0 = clock64();
#pragma unroll 32
    for (int i = 0; i < N / 4; i++) { s = __builtin_amdgcn_readfirstlane(v); asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "s"(s)); }
    c1 = clock64(); r1 = wall_clock64();
    if (lane == 0) { out[6] = c1 - c0; out[7] = r1 - r0; }
    // (e) independent VALU stream (issue rate)
    int w0 = v, w1 = v + 1, w2 = v + 2, w3 = v + 3;
    r0 = wall_clock64(); c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N / 4; i++) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(w0) : "v"(lane));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(w1) : "v"(lane));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(w2) : "v"(lane));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(w3) : "v"(lane));
    }
    c1 = clock64(); r1 = wall_clock64();
    if (lane == 0) { out[8] = c1 - c0; out[9] = r1 - r0; }
    // (f) taken branches
    r0 = wall_clock64(); c0 = clock64();
    for (int i = 0; i < N / 4; i++) { asm volatile("s_mul_i32 %0, %0, 5" : "+s"(s)); if (s == 0x7fffffff) break; } /* not unrolled: one taken branch per trip */
    c1 = clock64(); r1 = wall_clock64();
    if (lane == 0) { out[10] = c1 - c0; out[11] = r1 - r0; out[12] = v + s + p + w0 + w1 + w2 + w3; }
}
int main()
{
    unsigned long long *d, h[16];
    hipMalloc(&d, sizeof h);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rep);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    }
    const char *names[] = {"dependent VALU", "dependent SALU", "dependent LDS read", "readfirstlane + VALU pair", "independent VALU", "loop trip (s_mul, s_cmp, taken branch)"};
    const int counts[] = {N, N, N / 4, N / 4, N, N / 4};
    for (int i = 0; i < 6; i++)
        printf("%-40s %7.2f shader cycles, %7.2f ns each (clock64 %llu, 100 MHz ticks %llu)\n", names[i], (double)h[2 * i] / counts[i],
               (double)h[2 * i + 1] * 10.0 / counts[i], h[2 * i], h[2 * i + 1]);
    return 0;
}
