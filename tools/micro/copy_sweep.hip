// copy-kernel sweep: what a plain 16-byte-per-lane copy reaches on this part (read + write bytes / time), by block size, loads in flight per
// thread, grid shape and buffer size.  hipcc --offload-arch=gfx950 -O3 -o copy_sweep copy_sweep.hip && ./copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void k_copy(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
template <int U, bool NT> double run(const v4u* a, v4u* b, size_t bytes, int block, int grid, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(block), 0, 0, a, b, bytes / 16);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(block), 0, 0, a, b, bytes / 16);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * bytes * reps / (ms * 1e-3) / 1e12;
}
int main()
{
    for (size_t gb : {1, 4}) {
        const size_t bytes = gb << 30;
        v4u *a, *b; (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMemset(a, 1, bytes);
        for (int block : {256, 512, 1024})
            for (int mult : {8, 32, 128}) {
                const int grid = 256 * mult * 256 / block;
                printf("%zu GB block %4d grid %6d | U1 %.2f U2 %.2f U4 %.2f U8 %.2f | nt U1 %.2f U4 %.2f TB/s\n", gb, block, grid,
                       run<1, false>(a, b, bytes, block, grid, 10), run<2, false>(a, b, bytes, block, grid, 10), run<4, false>(a, b, bytes, block, grid, 10),
                       run<8, false>(a, b, bytes, block, grid, 10), run<1, true>(a, b, bytes, block, grid, 10), run<4, true>(a, b, bytes, block, grid, 10));
            }
        {   // one thread per element, no loop
            const size_t n16 = bytes / 16;
            const int block = 256; const size_t grid = n16 / block;
            printf("%zu GB direct grid %zu: %.2f TB/s; hipMemcpyDtoD: ", gb, grid, run<1, false>(a, b, bytes, block, (int)grid, 10));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
            hipEventRecord(e0, 0); for (int r = 0; r < 10; ++r) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); printf("%.2f TB/s\n", 2.0 * bytes * 10 / (ms * 1e-3) / 1e12);
        }
        hipFree(a); hipFree(b);
    }
}
