// What a hand-off between two waves costs on gfx950: two single-wave workgroups pass a counter back and forth through
// global memory (store + polling load, both at the given scope), 2000 round trips; the two workgroups are picked by their
// XCC_ID so that they sit on the SAME XCD or on DIFFERENT XCDs.  Also: the latency of one load of a line another wave
// has just written.  The cost model behind the row kernels (VP8) and the done flags (HEVC): DESIGN.md section 4.7.
// Build: hipcc --offload-arch=gfx950 -O3 -o handoff.bin microbench_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define ROUNDS 2000
struct Ctl { unsigned ball; unsigned pad0[31]; unsigned claim[2]; unsigned pad1[30]; unsigned long long ticks[2]; unsigned xcc[2]; };
template <int SCOPE> /* 1: agent (sc1), 2: system (sc0 sc1) */
__device__ __forceinline__ unsigned ld(const unsigned *p)
{
    return SCOPE == 1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int SCOPE>
__device__ __forceinline__ void st(unsigned *p, unsigned v)
{
    if (SCOPE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// want_same: both players must report the same XCC_ID; else different ones.  The first workgroup claims role 0; a later one
// with a fitting XCC_ID claims role 1; everybody else leaves.
template <int SCOPE>
__global__ __launch_bounds__(64) void k(Ctl *c, int want_same)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    int role = -1;
    if (threadIdx.x == 0) {
        if (atomicCAS(&c->claim[0], 0u, xcc + 1) == 0u) role = 0;
        else {
            unsigned first;
            while ((first = __hip_atomic_load(&c->claim[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {}
            const bool same = first - 1 == xcc;
            if (same == (want_same != 0) && atomicCAS(&c->claim[1], 0u, xcc + 1) == 0u) role = 1;
        }
    }
    role = __builtin_amdgcn_readfirstlane(role);
    if (role < 0) return;
    if (threadIdx.x == 0) c->xcc[role] = xcc;
    // wait for the partner (bounded: it may never come if no workgroup fits)
    for (int spins = 0; __hip_atomic_load(&c->claim[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u; spins++)
        if (spins > (1 << 22)) return;
    const unsigned long long t0 = wall_clock64();
    for (unsigned r = 0; r < ROUNDS; r++) {
        const unsigned mine = 2 * r + (unsigned)role + 1; /* role 0 waits for an even ball and makes it odd, role 1 the other way */
        int spins = 0;
        while (ld<SCOPE>(&c->ball) != mine - 1) { if (++spins > (1 << 22)) return; }
        if (threadIdx.x == 0) st<SCOPE>(&c->ball, mine);
    }
    if (threadIdx.x == 0) c->ticks[role] = wall_clock64() - t0;
}
int main()
{
    Ctl *c;
    hipMalloc(&c, sizeof(Ctl));
    for (int scope = 1; scope <= 2; scope++)
        for (int same = 1; same >= 0; same--) {
            Ctl h;
            memset(&h, 0, sizeof h);
            hipMemcpy(c, &h, sizeof h, hipMemcpyHostToDevice);
            if (scope == 1) hipLaunchKernelGGL(k<1>, dim3(64), dim3(64), 0, 0, c, same);
            else hipLaunchKernelGGL(k<2>, dim3(64), dim3(64), 0, 0, c, same);
            hipDeviceSynchronize();
            hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost);
            const double us = (double)h.ticks[0] / 100.0; /* wall_clock64: 100 MHz */
            printf("scope %s, players on %s XCD (XCC_ID %u and %u): %.3f us per round trip (two hand-offs), %.3f us per hand-off\n",
                   scope == 1 ? "agent (sc1)" : "system (sc0 sc1)", same ? "the SAME" : "DIFFERENT", h.xcc[0], h.xcc[1], us / ROUNDS, us / ROUNDS / 2);
        }
    return 0;
}
