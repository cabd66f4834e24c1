// Which workgroups share an XCD, and what does a flag hand-over between two CUs cost at each memory scope?
// hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip && ./xcd_pingpong
// Every spin loop is bounded (a hand-over that never becomes visible ends the test, it cannot hang the GPU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ __launch_bounds__(64) void k_xcc(int* out)
{
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf);
}

// SCOPE: 1 workgroup, 2 agent, 3 system.  Block A and block B bounce a counter: A waits for even->odd by B, etc.
template <int SCOPE>
__device__ __forceinline__ int ld(int* p)
{
    if (SCOPE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (SCOPE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int SCOPE>
__device__ __forceinline__ void st(int* p, int v)
{
    if (SCOPE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (SCOPE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int SCOPE>
__global__ __launch_bounds__(64) void k_pingpong(int* flag, int a, int b, int rounds, long long* cycles, int* done)
{
    const int me = blockIdx.x == a ? 0 : blockIdx.x == b ? 1 : -1;
    if (me < 0 || threadIdx.x != 0) return;
    const long long t0 = __builtin_readcyclecounter();
    int ok = 1;
    for (int r = 0; r < rounds && ok; ++r) {
        const int want = 2 * r + me;                // A waits for 2r, B for 2r+1
        int spins = 0;
        while (ld<SCOPE>(flag) != want) { if (++spins > 2000000) { ok = 0; break; } }
        if (ok) st<SCOPE>(flag, want + 1);
    }
    if (me == 0) { cycles[0] = __builtin_readcyclecounter() - t0; done[0] = ok; }
}

template <int SCOPE>
static void run(const char* name, int a, int b, int nblocks)
{
    int *flag, *done; long long* cyc;
    hipMalloc(&flag, 256); hipMalloc(&done, 4); hipMalloc(&cyc, 8);
    hipMemset(flag, 0, 256); hipMemset(done, 0, 4); hipMemset(cyc, 0, 8);
    const int rounds = 2000;
    hipLaunchKernelGGL(k_pingpong<SCOPE>, dim3(nblocks), dim3(64), 0, 0, flag, a, b, rounds, cyc, done);
    hipDeviceSynchronize();
    int d = 0; long long c = 0;
    hipMemcpy(&d, done, 4, hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s blocks %2d <-> %2d: %s, %.0f cycles per hand-over (one direction)\n", name, a, b, d ? "completed" : "NOT VISIBLE (gave up)", d ? (double)c / (2.0 * rounds) : 0.0);
    hipFree(flag); hipFree(done); hipFree(cyc);
}

int main()
{
    const int nb = 32;
    int* d; hipMalloc(&d, nb * 4);
    hipLaunchKernelGGL(k_xcc, dim3(nb), dim3(64), 0, 0, d);
    int h[nb]; hipMemcpy(h, d, nb * 4, hipMemcpyDeviceToHost);
    printf("XCC id of workgroups 0..%d:", nb - 1);
    for (int i = 0; i < nb; ++i) printf(" %d", h[i]);
    printf("\n");
    run<1>("workgroup", 0, 8, nb); run<2>("agent", 0, 8, nb); run<3>("system", 0, 8, nb);
    run<1>("workgroup", 0, 1, nb); run<2>("agent", 0, 1, nb); run<3>("system", 0, 1, nb);
    return 0;
}
