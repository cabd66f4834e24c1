// How many 256-thread "guest" workgroups does a CU take beside 24 resident one-wave "tenants" (the growth agents' footprint: 64 threads, 5 KB of LDS,
// 64 VGPRs)?  The guest's LDS size and VGPR allocation are parameters; it counts the guests co-resident on its CU (HW_ID / XCC_ID) while it spins.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/shadow tools/micro/shadow_occupancy.hip && /tmp/shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned cu_slot()
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((xcc & 15u) << 8) | ((hw >> 8) & 255u);      // xcc | se, sh, cu
}

__global__ __launch_bounds__(64) void k_tenant(long long ticks, int* sink)
{
    __shared__ int lds[1280];
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) { acc += lds[(threadIdx.x + acc) & 1023]; __builtin_amdgcn_s_sleep(8); }
    if (acc == 123456789) sink[0] = acc;
}

template <int VG>
__global__ __launch_bounds__(256) void k_guest(long long ticks, int* cnt, int* mx, int* sink)
{
    extern __shared__ int dyn[];
    if (VG >= 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    else if (VG >= 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    else asm volatile("v_mov_b32 v61, 0" ::: "v61");
    dyn[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned cu = cu_slot();
    if (threadIdx.x == 0) { const int now = atomicAdd(&cnt[cu], 1) + 1; atomicMax(&mx[cu], now); }
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) acc += dyn[(threadIdx.x + acc) & 255];
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(&cnt[cu], 1);
    if (acc == 123456789) sink[0] = acc;
}

template <int VG> static void run(int lds, bool withTenants, hipStream_t sa, hipStream_t sb, int* cnt, int* mx, int* sink)
{
    CK(hipMemset(cnt, 0, 4096 * 4)); CK(hipMemset(mx, 0, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (withTenants) hipLaunchKernelGGL(k_tenant, dim3(6144), dim3(64), 0, sa, 400000ll, sink);      // wall_clock64 ticks at 100 MHz: 4 ms
    CK(hipEventRecord(e0, sb));
    hipLaunchKernelGGL((k_guest<VG>), dim3(8192), dim3(256), lds, sb, 2000ll, cnt, mx, sink);             // 20 us per block
    CK(hipEventRecord(e1, sb));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int> h(4096); CK(hipMemcpy(h.data(), mx, 4096 * 4, hipMemcpyDeviceToHost));
    int cus = 0, hi = 0; long long sum = 0; std::vector<int> v;
    for (int x : h) if (x) { ++cus; hi = std::max(hi, x); sum += x; v.push_back(x); }
    std::sort(v.begin(), v.end());
    printf("guest LDS %5d B, %3d VGPRs, %s tenants: %7.3f ms for 8192 blocks; CUs seen %d, max co-resident guests per CU: median %d, max %d, mean %.2f\n",
           lds, VG, withTenants ? "with" : "no  ", ms, cus, v.empty() ? 0 : v[v.size() / 2], hi, cus ? (double)sum / cus : 0.0);
}

int main()
{
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    int *cnt, *mx, *sink; CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&mx, 4096 * 4)); CK(hipMalloc(&sink, 64));
    for (int rep = 0; rep < 2; ++rep)
        for (int t = 0; t < 2; ++t) {
            run<64>(17464, t, sa, sb, cnt, mx, sink);
            run<72>(25656, t, sa, sb, cnt, mx, sink);
            run<64>(8192, t, sa, sb, cnt, mx, sink);
            run<96>(8192, t, sa, sb, cnt, mx, sink);
        }
    return 0;
}
