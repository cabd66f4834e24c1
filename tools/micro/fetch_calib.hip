// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this part against kernels whose byte counts are known (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").  Every kernel touches a 4 GiB buffer (16 x the
// Infinity Cache) exactly once:  hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip;  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib
// and again with --pmc WRITE_SIZE; tools/micro/fetch_calib.py turns the two counter files into bytes-counted / bytes-known per kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k_read16(const v4u* __restrict__ p, size_t n, unsigned* sink) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { const v4u v = p[i]; if (v.x == 0x12345678u) sink[0] = v.y; } }
__global__ void k_read4(const unsigned* __restrict__ p, size_t n, unsigned* sink) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { const unsigned v = p[i]; if (v == 0x12345678u) sink[0] = v; } }
__global__ void k_read1(const uint8_t* __restrict__ p, size_t n, unsigned* sink) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { const unsigned v = p[i]; if (v == 0x77u) sink[0] = v; } }
// one 4-byte word per lane at a scattered address: every access falls into its own 128-byte line (stride 128 B, lines visited in a scrambled order)
__global__ void k_gather4(const unsigned* __restrict__ p, size_t nlines, unsigned* sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nlines) return;
    const size_t line = (i * 2654435761ull) % nlines;        // nlines is a power of two times an odd number? -- any bijection-ish scramble: collisions only lower the count
    const unsigned v = p[line * 32 + (i & 31)];
    if (v == 0x12345678u) sink[0] = v;
}
__global__ void k_write16(v4u* p, size_t n) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v4u{1u, 2u, 3u, (unsigned)i}; }
__global__ void k_write4(unsigned* p, size_t n) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
__global__ void k_scatter4(unsigned* p, size_t nlines) { const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < nlines) p[((i * 2654435761ull) % nlines) * 32 + (i & 31)] = (unsigned)i; }
int main()
{
    const size_t bytes = (size_t)4 << 30;
    void* buf; unsigned* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    (void)hipDeviceSynchronize();
    const size_t n16 = bytes / 16, n4 = bytes / 4, n1 = bytes / 4 /* 1 GiB of bytes */, nl = bytes / 128;
    hipLaunchKernelGGL(k_read16, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const v4u*)buf, n16, sink);
    hipLaunchKernelGGL(k_read4, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (const unsigned*)buf, n4, sink);
    hipLaunchKernelGGL(k_read1, dim3((unsigned)(n1 / 256)), dim3(256), 0, 0, (const uint8_t*)buf, n1, sink);
    hipLaunchKernelGGL(k_gather4, dim3((unsigned)(nl / 256)), dim3(256), 0, 0, (const unsigned*)buf, nl, sink);
    hipLaunchKernelGGL(k_write16, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (v4u*)buf, n16);
    hipLaunchKernelGGL(k_write4, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, (unsigned*)buf, n4);
    hipLaunchKernelGGL(k_scatter4, dim3((unsigned)(nl / 256)), dim3(256), 0, 0, (unsigned*)buf, nl);
    (void)hipDeviceSynchronize();
    printf("known bytes: read16 %zu read4 %zu read1 %zu gather4 %zu words (%zu lines of 128 B) write16 %zu write4 %zu scatter4 %zu words\n", bytes, bytes, n1, nl, nl, bytes, bytes, nl);
    return 0;
}
