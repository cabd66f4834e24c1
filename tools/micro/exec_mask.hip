// Does a CDNA4 SIMD spend less pipe time on a wave64 VALU instruction when only part of EXEC is set?
// hipcc --offload-arch=gfx950 -O3 -o exec_mask exec_mask.hip && ./exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int LANES>
__global__ __launch_bounds__(64) void k(double* out, int iters, double seed)
{
    const int lane = threadIdx.x;
    double a = seed + lane, b = 1.0000001, c = 0.5;
    float fa = (float)seed + lane, fb = 1.0001f, fc = 0.25f;
    if (lane < LANES) {
        for (int i = 0; i < iters; ++i) {
            if (MODE == 0) {   // dependent fp64 add / mul
#pragma unroll
                for (int u = 0; u < 16; ++u) { a = __dadd_rn(a, c); a = __dmul_rn(a, b); }
            } else if (MODE == 1) {   // dependent fp32 fma
#pragma unroll
                for (int u = 0; u < 32; ++u) fa = __fmaf_rn(fa, fb, fc);
            } else if (MODE == 2) {   // cvt round trip
#pragma unroll
                for (int u = 0; u < 16; ++u) { fa = (float)__dadd_rn((double)fa, c); }
            } else {            // rcp
#pragma unroll
                for (int u = 0; u < 16; ++u) { fa = __builtin_amdgcn_rcpf(fa) + fc; }
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a + fa;
}

template <int MODE, int LANES>
static void run(const char* name, int waves_per_simd, int instr_per_iter)
{
    const int blocks = 1024 * waves_per_simd, iters = 20000;
    double* d; hipMalloc(&d, (size_t)blocks * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, LANES>), dim3(blocks), dim3(64), 0, 0, d, 100, 1.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, LANES>), dim3(blocks), dim3(64), 0, 0, d, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * iters * instr_per_iter;
    printf("%-28s lanes %2d waves/SIMD %d: %.2f ms, %.2f cycles/instr/SIMD at 2.4 GHz\n", name, LANES, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(d);
}

int main()
{
    for (int w : {1, 4, 8}) {
        if (w == 1) { run<0, 64>("fp64 add+mul dependent", 1, 32); run<0, 32>("fp64 add+mul dependent", 1, 32); run<0, 1>("fp64 add+mul dependent", 1, 32); }
        if (w == 4) { run<0, 64>("fp64 add+mul dependent", 4, 32); run<0, 32>("fp64 add+mul dependent", 4, 32); run<0, 1>("fp64 add+mul dependent", 4, 32); }
        if (w == 8) { run<0, 64>("fp64 add+mul dependent", 8, 32); run<0, 32>("fp64 add+mul dependent", 8, 32); run<0, 1>("fp64 add+mul dependent", 8, 32); }
        if (w == 1) { run<1, 64>("fp32 fma dependent", 1, 32); run<1, 32>("fp32 fma dependent", 1, 32); run<1, 1>("fp32 fma dependent", 1, 32); }
        if (w == 4) { run<1, 64>("fp32 fma dependent", 4, 32); run<1, 32>("fp32 fma dependent", 4, 32); run<1, 1>("fp32 fma dependent", 4, 32); }
        if (w == 8) { run<1, 64>("fp32 fma dependent", 8, 32); run<1, 32>("fp32 fma dependent", 8, 32); run<1, 1>("fp32 fma dependent", 8, 32); }
        if (w == 8) { run<2, 64>("cvt+fp64 add+cvt", 8, 48); run<2, 32>("cvt+fp64 add+cvt", 8, 48); run<2, 1>("cvt+fp64 add+cvt", 8, 48); }
        if (w == 8) { run<3, 64>("rcp+add", 8, 32); run<3, 32>("rcp+add", 8, 32); run<3, 1>("rcp+add", 8, 32); }
    }
    return 0;
}
