// Litmus test of the ordering the multi-group growth relies on (csrc/lsd_grow.hip, rel_fence<true> / mw_commit_mg): message passing between two workgroups on
// DIFFERENT CUs -- same XCD and different XCDs -- with relaxed agent-scope accesses ordered by `s_waitcnt vmcnt(0)` only (no release / acquire fence, no cache
// write-back or invalidate):
//     writer:  X <- r (relaxed agent-scope store, or atomicMin as the owner words / notice words use)  ;  s_waitcnt vmcnt(0)  ;  Y <- r (relaxed agent-scope store)
//     reader:  y <- Y (relaxed agent-scope load)  ;  s_waitcnt vmcnt(0)  ;  x <- X (relaxed agent-scope load)         -- read in the opposite order
// The protocol is correct iff the reader never sees x older than y (the notice is performed before the watermark that passes it; a commit wave that has seen the
// watermark sees the notice).  A violation is counted, never waited for; every loop is bounded.  A third set of workgroups hammers the same cache lines' neighbours
// (NOISE) so that the lines move between the L2s while the test runs.
//   hipcc --offload-arch=gfx950 -O3 -o mp_litmus mp_litmus.hip && ./mp_litmus [rounds]
// Prints one JSON line: for every (placement, variant) the number of reader observations and of violations; exit code 1 if any violation was seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__) || defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__)
// (gfx9: stores and atomics without return count in vmcnt -- the premise of the protocol; gfx10+ counts them in vscnt)
#else
#error "mp_litmus.hip tests a gfx9 property (stores counted in vmcnt)"
#endif

__global__ __launch_bounds__(64) void k_xcc(int* out)
{
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf);
}

__device__ __forceinline__ unsigned ag_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// VARIANT 0: X written by a relaxed store; 1: X lowered by atomicMin (X counts DOWN from 0xffffffff - the owner-word form); 2: no s_waitcnt on either side (the
// control: shows whether the test can see a reordering at all on this part)
template <int VARIANT>
__global__ __launch_bounds__(64) void k_mp(unsigned* mem, int wblock, int rblock, int rounds, unsigned long long* res)
{
    unsigned* X = mem;             // one cache line each
    unsigned* Y = mem + 64;
    unsigned* STOP = mem + 128;
    unsigned* noise = mem + 192;
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (b == wblock) {
        for (int r = 1; r <= rounds; ++r) {
            if (VARIANT == 1) __hip_atomic_fetch_min(X, 0xffffffffu - (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ag_store(X, (unsigned)r);
            if (VARIANT != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ag_store(Y, (unsigned)r);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ag_store(STOP, 1u);
    } else if (b == rblock) {
        unsigned long long seen = 0, bad = 0, distinct = 0;
        unsigned lasty = 0;
        for (long long it = 0; it < 40ll * rounds; ++it) {
            const unsigned y = ag_load(Y);
            if (VARIANT != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("" ::: "memory");
            unsigned x = ag_load(X);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (VARIANT == 1) x = 0xffffffffu - x;          // back to "rounds performed"
            ++seen;
            if (x < y) ++bad;                                // the flag was seen, the data it announces was not
            if (y != lasty) { ++distinct; lasty = y; }
            if (ag_load(STOP) && y == (unsigned)rounds) break;
        }
        res[0] = seen; res[1] = bad; res[2] = distinct;
    } else {
        // noise: read-modify-write traffic on the neighbouring lines from every other CU until the writer is done (bounded)
        for (int it = 0; it < 4 * rounds; ++it) {
            __hip_atomic_fetch_add(noise + 64 * (b & 15), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((it & 63) == 0 && ag_load(STOP)) break;
        }
    }
}

template <int VARIANT>
static void run(const char* name, const char* place, int wb, int rb, int nblocks, int rounds, bool* any_bad, bool control)
{
    unsigned* mem; unsigned long long* res;
    hipMalloc(&mem, 64 * 4 * (4 + 16)); hipMalloc(&res, 24);
    hipMemset(mem, 0, 64 * 4 * (4 + 16)); hipMemset(res, 0, 24);
    if (VARIANT == 1) { unsigned ff = 0xffffffffu; hipMemcpy(mem, &ff, 4, hipMemcpyHostToDevice); }
    hipLaunchKernelGGL(k_mp<VARIANT>, dim3(nblocks), dim3(64), 0, 0, mem, wb, rb, rounds, res);
    hipDeviceSynchronize();
    unsigned long long h[3] = {0, 0, 0};
    hipMemcpy(h, res, 24, hipMemcpyDeviceToHost);
    printf("  {\"variant\": \"%s\", \"placement\": \"%s\", \"writer_block\": %d, \"reader_block\": %d, \"observations\": %llu, \"distinct_flags_seen\": %llu, \"violations\": %llu},\n",
           name, place, wb, rb, h[0], h[2], h[1]);
    if (h[1] && !control) *any_bad = true;
    hipFree(mem); hipFree(res);
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000000;
    const int nb = 64;
    int* d; hipMalloc(&d, nb * 4);
    hipLaunchKernelGGL(k_xcc, dim3(nb), dim3(64), 0, 0, d);
    std::vector<int> xcc(nb);
    hipMemcpy(xcc.data(), d, nb * 4, hipMemcpyDeviceToHost);
    // a pair of blocks on one XCD and a pair on two different XCDs, from the placement this launch shape really gets
    int same = -1, other = -1;
    for (int i = 1; i < nb && (same < 0 || other < 0); ++i) {
        if (same < 0 && xcc[i] == xcc[0]) same = i;
        if (other < 0 && xcc[i] != xcc[0]) other = i;
    }
    bool bad = false;
    printf("{\"rounds\": %d, \"xcc_of_blocks_0_15\": [", rounds);
    for (int i = 0; i < 16; ++i) printf("%d%s", xcc[i], i < 15 ? ", " : "");
    printf("], \"results\": [\n");
    if (other > 0) {
        run<0>("store ; s_waitcnt vmcnt(0) ; store", "different XCDs", 0, other, nb, rounds, &bad, false);
        run<1>("atomicMin ; s_waitcnt vmcnt(0) ; store", "different XCDs", 0, other, nb, rounds, &bad, false);
        run<0>("store ; s_waitcnt vmcnt(0) ; store", "different XCDs (roles swapped)", other, 0, nb, rounds, &bad, false);
        run<2>("control: no s_waitcnt", "different XCDs", 0, other, nb, rounds, &bad, true);
    }
    if (same > 0) {
        run<0>("store ; s_waitcnt vmcnt(0) ; store", "one XCD, two CUs", 0, same, nb, rounds, &bad, false);
        run<1>("atomicMin ; s_waitcnt vmcnt(0) ; store", "one XCD, two CUs", 0, same, nb, rounds, &bad, false);
        run<2>("control: no s_waitcnt", "one XCD, two CUs", 0, same, nb, rounds, &bad, true);
    }
    printf("  {}], \"ordering_holds\": %s}\n", bad ? "false" : "true");
    return bad ? 1 : 0;
}
