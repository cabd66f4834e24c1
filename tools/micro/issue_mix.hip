// How many instructions per cycle does a CDNA4 SIMD issue for the instruction mixes of the one-wave agents (lsd.hip k_lsd_grow, lsd_seedsort.hip)?
// One-wave workgroups, W waves per SIMD, tight loops written in inline assembly:
//   V   : 32 v_fma_f32 in 4 independent chains           (vector only)
//   VD  : 32 v_fma_f32 in ONE dependent chain
//   S   : 32 s_add_u32 in 4 independent chains           (scalar only)
//   VS  : 16 v_fma_f32 + 16 s_add_u32 interleaved, independent of each other   (do scalar instructions issue beside vector ones, or instead of them?)
//   VSD : v_readfirstlane -> s_add -> v_add (sgpr operand) -> ... one dependent chain through both files (the agents' accept chain)
//   V64 : 32 v_fma_f64 in 4 chains
// Prints wave-instructions per cycle per SIMD (2.4 GHz nominal; the clock under load is lower, so ratios between rows are what matters).
// hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float b = 1.0001f, c = 0.25f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const double db = 1.0000001, dc = 0.5;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 %4, %4, 3\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 %5, %5, 5\n"
                             "v_fma_f32 %2, %2, %8, %9\n s_add_u32 %6, %6, 7\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 %7, %7, 9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b), "v"(c) : "scc");
        } else if (MODE == 4) {
            int vi = __float_as_int(a0);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 3\n v_add_u32 %0, %1, %0\n v_xor_b32 %0, 1, %0" : "+v"(vi), "+s"(s0) :: "scc");
            a0 = __int_as_float(vi);
        } else if (MODE == 5) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));
        } else if (MODE == 6) {      // a vector compare into a scalar pair, a scalar logic op on it, a branch-free select: the agents' mask arithmetic
            int vi = __float_as_int(a0);
            unsigned long long m = s0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_cmp_gt_u32 %1, %0, %2\n s_and_b64 %1, %1, exec\n v_cndmask_b32 %0, %0, %2, %1\n v_add_u32 %0, 1, %0" : "+v"(vi), "+s"(m) : "v"(threadIdx.x) : "scc");
            a0 = __int_as_float(vi); s0 = (unsigned)m;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3) + (float)(s0 + s1 + s2 + s3);
}

template <int MODE>
static void run(const char* name, int instr_per_iter)
{
    for (int w : {1, 2, 4, 6, 8}) {
        const int blocks = 1024 * w, iters = 20000;
        float* d; hipMalloc(&d, (size_t)blocks * 64 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, 200);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)w * iters * instr_per_iter;
        printf("%-44s waves/SIMD %d: %7.2f ms  %.3f instr/cycle/SIMD  (%.2f cycles per instruction)\n", name, w, ms, n / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / n);
        hipFree(d);
    }
}

int main()
{
    run<0>("V   32 v_fma_f32, 4 chains", 32);
    run<1>("VD  32 v_fma_f32, 1 chain", 32);
    run<2>("S   32 s_add_u32, 4 chains", 32);
    run<3>("VS  16 v_fma + 16 s_add interleaved", 32);
    run<4>("VSD readfirstlane/s_add/v_add/v_xor chain", 32);
    run<5>("V64 32 v_fma_f64, 4 chains", 32);
    run<6>("M   v_cmp/s_and/v_cndmask/v_add chain", 32);
    return 0;
}
