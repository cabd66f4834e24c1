// lsd_sort.hip -- the pseudo-ordering of cv::LineSegmentDetector's seeds (OpenCV 3.4 lsd.cpp ll_angle: pixels bucketed by gradient-norm bin,
// the buckets visited from the highest bin down; restated in oracle/line_oracle.cpp:96-116, convention C.9 variant 0).
//
// k_lsd_keys (lsd.hip) emits the keys of an image's defined pixels in raster order: (n_bins - 1 - bin) << 22 | pixel address.  The seed order
// is "field ascending, raster order inside a field value", i.e. a STABLE sort by the 10-bit field alone: a least-significant-digit radix
// sort with two 5-bit digits, one segment per image.  Per digit:
//   k_radix_hist     per 8192-key chunk: how many keys carry each of the 32 digit values                  -> hist[image][chunk][32]
//   k_radix_scan     per image: exclusive prefix over the chunks per digit value, exclusive scan of the totals over the 32 values
//   k_radix_scatter  per chunk: the waves take consecutive parts; a wave walks its quarter 64 keys at a time, ranks equal digits
//                    inside the step by lane order (match-any from 5 ballots) and keeps the running bucket positions in LDS
// 32 buckets, not 1024 in one pass: a chunk then owns runs of ~256 consecutive keys per bucket; it orders its keys in LDS first and writes
// each run with consecutive lanes.  (Measured on the way: a single-pass 1024-bucket scatter -- runs of 8 keys = half a 64-byte sector, 16
// chunks of an image interleaving in every bucket -- 9.7 ms for the scatter alone; two 32-bucket passes storing straight from the ranking
// loop 4.6 + 3.1 ms; the library sort this file replaces, rocPRIM segmented_radix_sort_keys, 5.9 ms.)
#include "lsd_device.hpp"

namespace olf {

constexpr int SORT_CHUNK = 8192, RB = 32;

// lanes of the wave whose digit equals this lane's (valid lanes only)
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool valid)
{
    unsigned long long m = wave_vote(valid);
#pragma unroll
    for (int bit = 0; bit < 5; ++bit) {
        const unsigned long long bm = wave_vote((d >> bit) & 1u);
        m &= ((d >> bit) & 1u) ? bm : ~bm;
    }
    return m;
}

template <int SHIFT>
__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t* __restrict__ keysAll, const int* __restrict__ keyCount, int Ps,
                                                    uint32_t* __restrict__ histAll, int maxChunks)
{
    __shared__ uint32_t h[RB];
    const int img = blockIdx.y, kc = blockIdx.x, lane = threadIdx.x & 63;
    const int nkeys = keyCount[img * 32], base = kc * SORT_CHUNK;
    if (base >= nkeys) return;
    if (threadIdx.x < RB) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* keys = keysAll + (size_t)img * Ps + base;
    const int n = min(SORT_CHUNK, nkeys - base);
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {      // 8 independent loads in flight per thread
        uint32_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = i0 + u * 256 < n ? keys[i0 + u * 256] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool valid = i0 + u * 256 < n;
            const uint32_t d = (k[u] >> SHIFT) & (RB - 1);
            const unsigned long long m = match_digit(d, valid);
            if (valid && wave_rank_below(m) == 0) atomicAdd(&h[d], (uint32_t)__popcll(m));      // one atomic per digit value and wave step (plain per-key LDS atomics: 0.95 vs 0.88 ms)
        }
    }
    __syncthreads();
    if (threadIdx.x < RB) histAll[((size_t)img * maxChunks + kc) * RB + threadIdx.x] = h[threadIdx.x];
}

// one wave per image; lane = digit value (lanes 32..63 idle)
__global__ __launch_bounds__(64) void k_radix_scan(uint32_t* __restrict__ histAll, const int* __restrict__ keyCount, uint32_t* __restrict__ baseAll,
                                                   int maxChunks)
{
    const int img = blockIdx.x, b = threadIdx.x & (RB - 1);
    const int nkeys = keyCount[img * 32];
    const int nkc = (nkeys + SORT_CHUNK - 1) / SORT_CHUNK;
    uint32_t* h = histAll + (size_t)img * maxChunks * RB + b;
    uint32_t run = 0;
    if (threadIdx.x < RB) {
        for (int c0 = 0; c0 < nkc; c0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = c0 + u < nkc ? h[(size_t)(c0 + u) * RB] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (c0 + u < nkc) { h[(size_t)(c0 + u) * RB] = run; run += v[u]; }
        }
    }
    uint32_t inc = threadIdx.x < RB ? run : 0u;
#pragma unroll
    for (int o = 1; o < RB; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += t; }
    if (threadIdx.x < RB) baseAll[(size_t)img * RB + b] = inc - run;
}

constexpr int SCATTER_THREADS = 512;      // per 8192-key chunk (256: 2.9 ms per pass, 512: 2.35, 1024: 2.75)
template <int SHIFT>
__global__ __launch_bounds__(SCATTER_THREADS) void k_radix_scatter(const uint32_t* __restrict__ keysAll, uint32_t* __restrict__ outAll, const int* __restrict__ keyCount,
                                                       int Ps, const uint32_t* __restrict__ histAll, const uint32_t* __restrict__ baseAll, int maxChunks)
{
    __shared__ uint32_t stage[SORT_CHUNK];      // the chunk's keys in bucket order: written to memory as full, coalesced runs
    constexpr int NWV = SCATTER_THREADS / 64;
    __shared__ uint32_t pos[NWV][RB];           // per wave: where (in `stage`) the next key of each digit value goes
    __shared__ uint32_t lstart[RB + 1];         // where each bucket starts in `stage`
    __shared__ uint32_t gstart[RB];             // ... and in the image's output
    const int img = blockIdx.y, kc = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nkeys = keyCount[img * 32], base = kc * SORT_CHUNK;
    if (base >= nkeys) return;
    const uint32_t* keys = keysAll + (size_t)img * Ps + base;
    uint32_t* out = outAll + (size_t)img * Ps;
    const int n = min(SORT_CHUNK, nkeys - base);
    constexpr int Q = SORT_CHUNK / NWV;
    const int low = wv * Q, hiw = min(n, (wv + 1) * Q);
    if (threadIdx.x < NWV * RB) (&pos[0][0])[threadIdx.x] = 0;
    __syncthreads();
    // the waves' own histograms
    for (int i0 = low + lane; i0 < hiw; i0 += 8 * 64) {
        uint32_t k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = i0 + u * 64 < hiw ? keys[i0 + u * 64] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool valid = i0 + u * 64 < hiw;
            const uint32_t d = (k[u] >> SHIFT) & (RB - 1);
            const unsigned long long m = match_digit(d, valid);
            if (valid && wave_rank_below(m) == 0) pos[wv][d] += (uint32_t)__popcll(m);      // this wave owns pos[wv] (per-key LDS atomics: 3.5 vs 2.9 ms)
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // bucket starts (chunk-local and global), then the waves' starting positions inside the buckets
        const int b = threadIdx.x & (RB - 1);
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) c += pos[w][b];
        uint32_t inc = threadIdx.x < RB ? c : 0u;
#pragma unroll
        for (int o = 1; o < RB; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += t; }
        if (threadIdx.x < RB) {
            uint32_t o = inc - c;
            lstart[b] = o;
            if (b == RB - 1) lstart[RB] = inc;
            gstart[b] = histAll[((size_t)img * maxChunks + kc) * RB + b] + baseAll[(size_t)img * RB + b];
#pragma unroll
            for (int w = 0; w < NWV; ++w) { const uint32_t t = pos[w][b]; pos[w][b] = o; o += t; }
        }
    }
    __syncthreads();
    uint32_t knext = low + lane < hiw ? keys[low + lane] : 0u;
    for (int i0 = low; i0 < hiw; i0 += 64) {
        const uint32_t key = knext;
        const bool valid = i0 + lane < hiw;
        knext = i0 + 64 + lane < hiw ? keys[i0 + 64 + lane] : 0u;
        const uint32_t d = (key >> SHIFT) & (RB - 1);
        const unsigned long long m = match_digit(d, valid);      // equal digits of this step, in lane order = input order: stable
        if (valid) {
            const int rank = wave_rank_below(m), cnt = __popcll(m);
            const uint32_t p = pos[wv][d] + (uint32_t)rank;
            stage[p] = key;
            if (rank == cnt - 1) pos[wv][d] = p + 1;
        }
        __builtin_amdgcn_wave_barrier();       // LDS operations of one wave execute in order; keep the compiler from moving the next step's read up
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += SCATTER_THREADS) {
        const uint32_t key = stage[i];
        const uint32_t d = (key >> SHIFT) & (RB - 1);
        out[gstart[d] + (uint32_t)i - lstart[d]] = key;
    }
}

int lsd_sort_max_chunks(int Ps) { return (Ps + SORT_CHUNK - 1) / SORT_CHUNK; }

// keysB (raster order, from k_lsd_keys) -> keysA (by the low digit) -> keysB (by the high digit): the order the growth kernels read
int launch_lsd_sort(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s)
{
    const int mc = lsd_sort_max_chunks(g.Ps);
    hipLaunchKernelGGL(k_radix_hist<22>, dim3(mc, n_images), dim3(256), 0, s, b.keysB, b.keyCount, g.Ps, b.sortHist, mc);
    hipLaunchKernelGGL(k_radix_scan, dim3(n_images), dim3(64), 0, s, b.sortHist, b.keyCount, b.sortBase, mc);
    hipLaunchKernelGGL(k_radix_scatter<22>, dim3(mc, n_images), dim3(SCATTER_THREADS), 0, s, b.keysB, b.keysA, b.keyCount, g.Ps, b.sortHist, b.sortBase, mc);
    hipLaunchKernelGGL(k_radix_hist<27>, dim3(mc, n_images), dim3(256), 0, s, b.keysA, b.keyCount, g.Ps, b.sortHist, mc);
    hipLaunchKernelGGL(k_radix_scan, dim3(n_images), dim3(64), 0, s, b.sortHist, b.keyCount, b.sortBase, mc);
    hipLaunchKernelGGL(k_radix_scatter<27>, dim3(mc, n_images), dim3(SCATTER_THREADS), 0, s, b.keysA, b.keysB, b.keyCount, g.Ps, b.sortHist, b.sortBase, mc);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
