// lsd_wide.hip -- the seed order of cv::LineSegmentDetector for the configurations whose sort key does not fit the 32-bit word of the fast path
// (lsd_seedsort.hip / lsd_sort.hip: (n_bins - 1 - bin) << 22 | pixel address): lsd_n_bins > 1024 or an LSD working image of 2^22 pixels and more
// (1080p at lsd_scale 2).  Both are free YAML keys of the reference (src/Config.cpp:268,274; Examples/PL/PL_KITTI00-02.yaml:110,116).  The key is the
// 64-bit word (n_bins - 1 - bin) << 32 | address here, and ONE kernel serves both conventions (C.9):
//   variant 1 (OpenCV >= 3.3): every pixel (x < w-1, y < h-1) in raster order through std::sort(begin, end, [](a, b) { return a.norm > b.norm; }) --
//             libstdc++'s introsort, replayed step by step with the comparison on the upper word only (oracle/line_oracle.cpp:443-453);
//   variant 0 (raster order inside a bin): the defined pixels' keys are distinct and ascend with the address, so the stable order IS the order of the
//             whole 64-bit word -- the same replay with the comparison on the whole word (FULL) sorts them; which unstable algorithm does not matter.
// This is the capacity path, not the fast one: a workgroup of 8 waves per image, no streaming tricks.
//   std::__sort             = __introsort_loop(first, last, 2 * floor(log2 n)) + __final_insertion_sort(first, last)
//   __introsort_loop        : while (last - first > 16) { depth_limit == 0 ? heap sort the range and stop : --depth_limit;
//                                 cut = __unguarded_partition_pivot(first, last); recurse on [cut, last); last = cut; }
//   __final_insertion_sort  : a stable sort of what the loop leaves; ranges are ordered among themselves, so every leaf (<= 16 elements) is sorted stably on its own.
// Ranges of more than WS_CAP elements are partitioned by the whole workgroup in global memory; smaller ones are queued, and when eight are waiting (or nothing
// else is left) every wave takes one into its slice of LDS and finishes its whole subtree there.  The Hoare partition in its rank formulation (lsd_seedsort.hip):
// the j-th element >= pivot from the left swaps with the j-th element <= pivot from the right while the former lies left of the latter; with J such pairs the
// cut is min(position of left stopper J, position of right stopper J - 1).  Stopper positions are compacted by a scan, J found by search (the predicate is
// monotone), the pairs swapped in parallel.  A range that can only hold undefined pixels (its lower key bound exceeds the smallest bin of a defined pixel) is
// left unsorted (variant 1): its elements never leave it and are never seeds.
// Output: the pixel addresses in seed order as 32-bit words (the growth agent's WIDE instantiation reads them unmasked) and their count.
#include "lsd_device.hpp"

namespace olf {

constexpr int WS_NT = 512, WS_NW = WS_NT / 64, WS_CAP = 512, WS_STK = 72, WS_WSTK = 64;

struct WsRange { int f, l, d; uint32_t lb; };

template <bool FULL> __device__ __forceinline__ unsigned long long wsK(unsigned long long e) { return FULL ? e : (e >> 32); }
__device__ __forceinline__ int wsU(int v) { return __builtin_amdgcn_readfirstlane(v); }

// std::__move_median_to_first(first, first + 1, mid, last - 1) (one lane)
template <bool FULL>
__device__ __forceinline__ void ws_median_to_first(unsigned long long* P, int first, int last)
{
    const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
    const unsigned long long ka = wsK<FULL>(P[a]), kb = wsK<FULL>(P[b]), kc = wsK<FULL>(P[c]);
    int m;
    if (ka < kb) { if (kb < kc) m = b; else if (ka < kc) m = c; else m = a; }
    else if (ka < kc) m = a;
    else if (kb < kc) m = c;
    else m = b;
    const unsigned long long t = P[first]; P[first] = P[m]; P[m] = t;
}

// libstdc++ __adjust_heap (+ __push_heap) and std::__partial_sort(first, last, last) = __make_heap + __sort_heap, one lane
template <bool FULL>
__device__ void ws_adjust_heap(unsigned long long* P, int first, int hole, int len, unsigned long long value)
{
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (wsK<FULL>(P[first + second]) < wsK<FULL>(P[first + second - 1])) --second;
        P[first + hole] = P[first + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        P[first + hole] = P[first + second - 1];
        hole = second - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && wsK<FULL>(P[first + parent]) < wsK<FULL>(value)) {
        P[first + hole] = P[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    P[first + hole] = value;
}
template <bool FULL>
__device__ void ws_heapsort(unsigned long long* P, int first, int last)
{
    const int len = last - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            ws_adjust_heap<FULL>(P, first, parent, len, P[first + parent]);
            if (parent == 0) break;
        }
    for (int l = last; l - first > 1;) {
        --l;
        const unsigned long long v = P[l];
        P[l] = P[first];
        ws_adjust_heap<FULL>(P, first, 0, l - first, v);
    }
}

// the Hoare partition of [f + 1, l) around the median moved to f, by the whole workgroup, in global memory; returns the cut, Kp = the pivot's key
template <bool FULL>
__device__ __forceinline__ int ws_partition_block(unsigned long long* K, uint32_t* posL, uint32_t* posR, int f, int l, unsigned long long& KpOut, int* s_cnt,
                                                  unsigned long long* s_b64, int* s_i)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { ws_median_to_first<FULL>(K, f, l); s_b64[0] = wsK<FULL>(K[f]); }
    __syncthreads();
    const unsigned long long Kp = s_b64[0];
    int nL = 0, nR = 0;
    for (int base = f + 1; base < l; base += WS_NT) {
        const int i = base + tid;
        const bool in = i < l;
        const unsigned long long k = in ? wsK<FULL>(K[i]) : 0ull;
        const bool isL = in && k >= Kp, isR = in && k <= Kp;
        const unsigned long long mL = wave_vote(isL), mR = wave_vote(isR);
        if (lane == 0) { s_cnt[wv] = (int)__popcll(mL); s_cnt[WS_NW + wv] = (int)__popcll(mR); }
        __syncthreads();
        int oL = nL, oR = nR, tL = 0, tR = 0;
#pragma unroll
        for (int w = 0; w < WS_NW; ++w) { const int a = s_cnt[w], b = s_cnt[WS_NW + w]; if (w < wv) { oL += a; oR += b; } tL += a; tR += b; }
        if (isL) posL[oL + wave_rank_below(mL)] = (uint32_t)i;
        if (isR) posR[oR + wave_rank_below(mR)] = (uint32_t)i;       // (ascending position: right stopper j from the right is posR[nR - 1 - j])
        nL += tL; nR += tR;
        __syncthreads();
    }
    nL = wsU(nL); nR = wsU(nR);
    // J = number of pairs that swap: posL[j] < posR[nR - 1 - j] holds for a prefix of j < min(nL, nR); 512-ary search
    int lo = 0, hi = min(nL, nR);
    while (hi > lo) {
        const int step = (hi - lo + WS_NT - 1) / WS_NT;
        const int j = lo + tid * step;
        const bool p = j < hi && posL[j] < posR[nR - 1 - j];
        const unsigned long long m = wave_vote(p);
        if (lane == 0) s_cnt[wv] = (int)__popcll(m);
        __syncthreads();
        int cnt = 0;
#pragma unroll
        for (int w = 0; w < WS_NW; ++w) cnt += s_cnt[w];
        cnt = wsU(cnt);
        __syncthreads();
        if (cnt == 0) hi = lo;
        else { const int nlo = lo + (cnt - 1) * step + 1, nhi = min(hi, lo + cnt * step); lo = nlo; hi = nhi; }
    }
    const int J = lo;
    for (int j = tid; j < J; j += WS_NT) {
        const uint32_t a = posL[j], b = posR[nR - 1 - j];
        const unsigned long long ea = K[a], eb = K[b];
        K[a] = eb; K[b] = ea;
    }
    if (tid == 0) {
        int cut = 0x7fffffff;
        if (J < nL) cut = (int)posL[J];
        if (J >= 1) cut = min(cut, (int)posR[nR - J]);
        s_i[0] = cut;
    }
    __syncthreads();
    KpOut = Kp;
    return wsU(s_i[0]);
}

// the same partition by one wave on its LDS copy of a range (positions relative to the copy)
template <bool FULL>
__device__ __forceinline__ int ws_partition_wave(unsigned long long* sb, unsigned short* pl, unsigned short* pr, int a, int b, unsigned long long& KpOut, int lane)
{
    if (lane == 0) ws_median_to_first<FULL>(sb, a, b);
    __builtin_amdgcn_wave_barrier();
    const unsigned long long Kp = wsK<FULL>(sb[a]);
    int nL = 0, nR = 0;
    for (int base = a + 1; base < b; base += 64) {
        const int i = base + lane;
        const bool in = i < b;
        const unsigned long long k = in ? wsK<FULL>(sb[i]) : 0ull;
        const bool isL = in && k >= Kp, isR = in && k <= Kp;
        const unsigned long long mL = wave_vote(isL), mR = wave_vote(isR);
        if (isL) pl[nL + wave_rank_below(mL)] = (unsigned short)i;
        if (isR) pr[nR + wave_rank_below(mR)] = (unsigned short)i;
        nL += (int)__popcll(mL); nR += (int)__popcll(mR);
    }
    __builtin_amdgcn_wave_barrier();
    const int M = min(nL, nR);
    int J = 0;
    for (int j0 = 0; j0 < M; j0 += 64) {
        const int j = j0 + lane;
        const unsigned long long m = wave_vote(j < M && pl[j] < pr[nR - 1 - j]);
        J += (int)__popcll(m);
        if (m != ~0ull) break;
    }
    for (int j = lane; j < J; j += 64) {
        const int x = pl[j], y = pr[nR - 1 - j];
        const unsigned long long ex = sb[x], ey = sb[y];
        sb[x] = ey; sb[y] = ex;
    }
    __builtin_amdgcn_wave_barrier();
    int cut = 0x7fffffff;
    if (J < nL) cut = pl[J];
    if (J >= 1) cut = min(cut, (int)pr[nR - J]);
    KpOut = Kp;
    return wsU(cut);
}

// one wave finishes the subtree of a range of <= WS_CAP elements in LDS: the rest of the introsort loop, then the final insertion sort of every leaf
template <bool FULL>
__device__ __forceinline__ void ws_small_range(unsigned long long* K, const WsRange r, unsigned long long Kthr, unsigned long long* sb, unsigned short* pl,
                                               unsigned short* pr, WsRange* stk, int lane)
{
    const int m = r.l - r.f;
    for (int i = lane; i < m; i += 64) sb[i] = K[r.f + i];
    __builtin_amdgcn_wave_barrier();
    int sp = 1;
    if (lane == 0) { WsRange t; t.f = 0; t.l = m; t.d = r.d; t.lb = r.lb; stk[0] = t; }
    __builtin_amdgcn_wave_barrier();
    while (sp > 0) {
        --sp;
        const WsRange t = stk[sp];
        int a = wsU(t.f), b = wsU(t.l), d = wsU(t.d);
        unsigned long long lb = (unsigned long long)(uint32_t)wsU((int)t.lb);
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            const int len = b - a;
            if (len <= 1) break;
            if (!FULL && lb > Kthr) break;                  // only undefined pixels: never seeds, never leave the range
            if (len <= 16) {
                // __final_insertion_sort on a leaf: stable by key
                const unsigned long long my = lane < len ? sb[a + lane] : 0ull, mk = wsK<FULL>(my);
                int rk = 0;
                for (int j = 0; j < len; ++j) { const unsigned long long kj = wsK<FULL>(sb[a + j]); rk += (kj < mk || (kj == mk && j < lane)) ? 1 : 0; }
                __builtin_amdgcn_wave_barrier();
                if (lane < len) sb[a + rk] = my;
                __builtin_amdgcn_wave_barrier();
                break;
            }
            if (d == 0) { if (lane == 0) ws_heapsort<FULL>(sb, a, b); __builtin_amdgcn_wave_barrier(); break; }
            --d;
            unsigned long long Kp;
            const int cut = ws_partition_wave<FULL>(sb, pl, pr, a, b, Kp, lane);
            if (lane == 0) { WsRange c; c.f = cut; c.l = b; c.d = d; c.lb = (uint32_t)Kp; stk[sp] = c; }      // (lb is only looked at when the key is the upper word)
            ++sp;
            __builtin_amdgcn_wave_barrier();
            b = cut;
        }
    }
    for (int i = lane; i < m; i += 64) K[r.f + i] = sb[i];
}

// nOverride / kthrOverride / depthOverride >= 0: debug entry (olf_debug_seed_sort_wide): n keys at keysA, list every key whose field is <= kthr, introsort's depth limit
template <bool FULL>
__global__ __launch_bounds__(WS_NT) void k_wide_sort(const LineGeom* __restrict__ gp, uint32_t* keysInAll, uint32_t* keysOutAll, int* __restrict__ keyCount,
                                                     const int* __restrict__ maxN, int* __restrict__ status, int nOverride, long long kthrOverride, int depthOverride)
{
    __shared__ unsigned long long s_sb[WS_NW][WS_CAP];
    __shared__ unsigned short s_pl[WS_NW][WS_CAP], s_pr[WS_NW][WS_CAP];
    __shared__ WsRange s_wstk[WS_NW][WS_WSTK];
    __shared__ WsRange s_stack[WS_STK], s_small[WS_NW];
    __shared__ int s_sp, s_nsmall, s_cnt[2 * WS_NW], s_i[4];
    __shared__ unsigned long long s_b64[2];
    const LineGeom& g = *gp;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t Ps = (size_t)g.Ps;
    unsigned long long* K = reinterpret_cast<unsigned long long*>(keysInAll) + (size_t)img * Ps;
    uint32_t* S = keysOutAll + (size_t)img * 2 * Ps;       // scratch while sorting (stopper positions), the seed list afterwards
    uint32_t* posL = S, * posR = S + Ps;
    int n = nOverride >= 0 ? nOverride : FULL ? keyCount[img * 32] : (g.Ws - 1) * (g.Hs - 1);
    unsigned long long Kthr = 0;
    bool empty = n <= 0;
    if (kthrOverride >= 0) Kthr = (unsigned long long)kthrOverride;
    else if (!empty) {
        const int mN = maxN[img * 32];
        if (mN <= 0) empty = true;                         // no defined pixel: no seed
        else {
            // the smallest bin a defined pixel can fall into, as k_lsd_keys bins it (lsd_seedsort.hip ss_sort_image)
            const double max_grad = sqrt((double)mN / 4.0);
            const double bin_coef = (double)(g.nBins - 1) / max_grad;
            const double normT = sqrt((double)g.nThr / 4.0);
            Kthr = (unsigned long long)(g.nBins - 1 - (int)(normT * bin_coef));
        }
    }
    if (empty) { if (tid == 0) keyCount[img * 32] = 0; return; }
    if (tid == 0) {
        int lg = 0; while ((2 << lg) <= n) ++lg;           // floor(log2 n)
        WsRange r; r.f = 0; r.l = n; r.d = depthOverride >= 0 ? depthOverride : 2 * lg; r.lb = 0u;
        s_stack[0] = r; s_sp = 1; s_nsmall = 0;
    }
    __syncthreads();
    for (;;) {
        const int sp = wsU(s_sp), ns = wsU(s_nsmall);
        if (ns == WS_NW || (sp == 0 && ns > 0)) {
            if (wv < ns) ws_small_range<FULL>(K, s_small[wv], Kthr, s_sb[wv], s_pl[wv], s_pr[wv], s_wstk[wv], lane);
            __syncthreads();
            if (tid == 0) s_nsmall = 0;
            __syncthreads();
            continue;
        }
        if (sp == 0) break;
        const WsRange r = s_stack[sp - 1];
        const int f = wsU(r.f), l = wsU(r.l), d = wsU(r.d);
        const uint32_t lb = (uint32_t)wsU((int)r.lb);
        __syncthreads();
        const bool drop = l - f <= 1 || (!FULL && (unsigned long long)lb > Kthr);
        if (drop || l - f <= WS_CAP) {
            if (tid == 0) { s_sp = sp - 1; if (!drop) { s_small[ns] = r; s_nsmall = ns + 1; } }
            __syncthreads();
            continue;
        }
        if (d == 0) {
            if (tid == 0) { ws_heapsort<FULL>(K, f, l); s_sp = sp - 1; }
            __syncthreads();
            continue;
        }
        unsigned long long Kp;
        const int cut = ws_partition_block<FULL>(K, posL, posR, f, l, Kp, s_cnt, s_b64, s_i);
        if (tid == 0) {
            if (sp + 1 > WS_STK) { atomicOr(status, 128); s_sp = 0; }      // (cannot happen: the stack holds one sibling per level of the depth limit)
            else {
                WsRange a; a.f = cut; a.l = l; a.d = d - 1; a.lb = (uint32_t)Kp;
                WsRange b; b.f = f; b.l = cut; b.d = d - 1; b.lb = lb;
                s_stack[sp - 1] = a; s_stack[sp] = b; s_sp = sp + 1;
            }
        }
        __syncthreads();
    }
    // the seed list: every key whose field is <= Kthr (a prefix: the array is sorted but for the ranges that hold larger fields only)
    if (tid == 0) {
        int cnt = n;
        if (!FULL || kthrOverride >= 0) {
            int lo = 0, hi = n;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((K[mid] >> 32) <= Kthr) lo = mid + 1; else hi = mid; }
            cnt = lo;
        }
        s_i[1] = cnt;
    }
    __syncthreads();
    const int cnt = s_i[1];
    for (int i = tid; i < cnt; i += WS_NT) S[i] = (uint32_t)K[i];
    if (tid == 0) keyCount[img * 32] = cnt;
}

int launch_lsd_sort_wide(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, long long kthrOverride, int depthOverride, int fullOverride)
{
    const bool full = fullOverride >= 0 ? fullOverride != 0 : g.seedOrder == 0;
    if (full)
        hipLaunchKernelGGL(k_wide_sort<true>, dim3(n_images), dim3(WS_NT), 0, s, b.geom, b.keysA, b.keysB, b.keyCount, b.maxN, b.status, nOverride, kthrOverride, depthOverride);
    else
        hipLaunchKernelGGL(k_wide_sort<false>, dim3(n_images), dim3(WS_NT), 0, s, b.geom, b.keysA, b.keysB, b.keyCount, b.maxN, b.status, nOverride, kthrOverride, depthOverride);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
