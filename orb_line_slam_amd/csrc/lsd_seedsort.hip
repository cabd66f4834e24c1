// lsd_seedsort.hip -- the seed order of cv::LineSegmentDetector as OpenCV >= 3.3 produces it (convention C.9, variant 1; restated with the real
// library call in oracle/line_oracle.cpp:111-121): every pixel (x < w-1, y < h-1) is pushed as {point, bin} in raster order and the vector goes
// through  std::sort(begin, end, [](a, b) { return a.norm > b.norm; })  -- an UNSTABLE sort, so the order of the seeds inside a gradient bin is
// whatever libstdc++'s introsort leaves.  That order is a pure function of the bin sequence, and this kernel replays it bit-exactly:
//
//   std::__sort            = __introsort_loop(first, last, 2 * floor(log2(n)))  +  __final_insertion_sort(first, last)
//   __introsort_loop       : while (last - first > 16) { depth_limit == 0 ? heap sort the range and stop : --depth_limit;
//                                cut = __unguarded_partition_pivot(first, last); recurse on [cut, last); last = cut; }
//   ..partition_pivot      : median of (first + 1, middle, last - 1) swapped to *first, then the Hoare partition of [first + 1, last) around it
//   __final_insertion_sort : a stable insertion sort, i.e. the final order = the order the loop leaves, stably sorted by key.
//
// Replay, one wave per image (a batch has thousands of images; inside an image the recursion is walked depth first, left to right, so finished
// pieces of the sorted list leave in list order):
//   * the Hoare partition of a range is order-isomorphic to "the j-th element >= pivot from the left swaps with the j-th element <= pivot from
//     the right while the former lies left of the latter".  The wave streams 64-element tiles from both ends, keeps each side's stoppers as a
//     lane mask, pairs them by rank through LDS and writes only the swapped elements back; the zone where the two scans meet lies inside the
//     last tile read and is resolved there (ss_zone).  Ranges of <= SS_CAP elements are copied to LDS once and never written back;
//   * a range whose keys are all equal (known from the pivots on the path to it) is a fixed permutation of its positions -- first <-> middle,
//     then [first + 1, last) reversed, cut in the middle, and so on down to the 16-element leaves: every element computes its final place
//     arithmetically (ss_emit_equal).  Four out of five small ranges are of this kind;
//   * a range that can only hold undefined pixels (all bins below the smallest bin a defined pixel can have) is dropped unsorted: its elements
//     never leave it and are never seeds.  81 % of the pixels of a typical image go this way after three or four levels;
//   * leaves (<= 16 elements) are ranked stably by key in registers and written to their place in the output;
//   * depth_limit == 0 (never on real images; tested with a forced limit): libstdc++'s heap sort (__make_heap + __sort_heap) replayed by one lane.
// Output: keysB = the keys ((n_bins - 1 - bin) << 22 | address) of all pixels whose bin is at least the smallest bin of a defined pixel, in
// seed order.  Undefined pixels that share that smallest bin are in the list too; the growth kernels skip them (NOTDEF bit of the gradient word).
#include "lsd_device.hpp"

namespace olf {

constexpr int SS_CAP = 1536;      // elements of a range held in LDS (6 KB: 24 waves = 24 images per CU)
constexpr int SS_PF = 4;          // tiles in flight per side while a range streams from memory

struct SsCtx {
    uint32_t* A;          // the image's keys in memory; sorted in place
    uint32_t* sbuf;       // LDS copy of [ldsFirst, ldsLast)
    uint32_t* xl;         // LDS, 64 words: stoppers of the left scan by rank
    uint32_t* xr;         // ... of the right scan
    uint32_t* out;        // the seed list
    int ldsFirst;
    int lane;
};

__device__ __forceinline__ uint32_t ssK(uint32_t e) { return e >> 22; }
template <bool LDS> __device__ __forceinline__ uint32_t ss_ld(const SsCtx& c, int i) { return LDS ? c.sbuf[i - c.ldsFirst] : c.A[i]; }
template <bool LDS> __device__ __forceinline__ void ss_st(const SsCtx& c, int i, uint32_t v) { if (LDS) c.sbuf[i - c.ldsFirst] = v; else c.A[i] = v; }
__device__ __forceinline__ int ss_rank_below(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// The meeting zone of the two scans, inside one tile (values v, lane l <-> position base + l).  GE: the zone's elements >= pivot (stoppers of the
// left scan, ranked from the left), LE: its elements <= pivot (stoppers of the right scan, ranked from the right).  Pair j swaps while
// L[j] < R[j]; the partition's return value is min(L[s], R[s - 1]) after s swaps, or the end of the zone if neither exists.
template <bool LDS>
__device__ __forceinline__ int ss_zone(const SsCtx& c, uint32_t& v, int base, unsigned long long GE, unsigned long long LE, int zoneEndLane)
{
    const int lane = c.lane;
    const bool isG = (GE >> lane) & 1ull, isLE = (LE >> lane) & 1ull;
    const int nG = __popcll(GE), nLE = __popcll(LE);
    const int rg = ss_rank_below(GE);
    const int rle = nLE - ss_rank_below(LE) - (isLE ? 1 : 0);
    if (isG) c.xl[rg] = (uint32_t)lane;
    if (isLE) c.xr[rle] = (uint32_t)lane;
    __builtin_amdgcn_wave_barrier();
    int partner = lane;
    bool swG = false, sw = false;
    if (isG && rg < nLE) { const int p = (int)c.xr[rg]; if (lane < p) { partner = p; swG = sw = true; } }
    if (isLE && rle < nG) { const int p = (int)c.xl[rle]; if (p < lane) { partner = p; sw = true; } }      // (an element equal to the pivot is in both sets but can only swap as one of them)
    __builtin_amdgcn_wave_barrier();
    const int s = __popcll(__ballot(swG));
    const uint32_t nv = (uint32_t)__shfl((int)v, partner);
    if (sw) { v = nv; ss_st<LDS>(c, base + lane, nv); }
    const unsigned long long cand = __ballot((isG && rg == s) || (isLE && s >= 1 && rle == s - 1));
    return base + (cand ? (int)__builtin_ctzll(cand) : zoneEndLane);
}

// std::__unguarded_partition(lo, hi, pivot) on [lo, hi) (= [first + 1, last)) with comp(a, b) = K(a) < K(b); returns the cut
template <bool LDS>
__device__ __forceinline__ int ss_partition(const SsCtx& c, int lo, int hi, uint32_t Kp)
{
    const int lane = c.lane;
    int lc = lo, rc = hi;                    // unread: [lc, rc)
    uint32_t vL = 0, vR = 0;                 // the current tile of either side; lane l <-> position baseL + l / baseR + l
    int baseL = 0, baseR = 0;
    bool validL = false, validR = false;
    unsigned long long LQ = 0, RQ = 0;       // stoppers of the current tiles that have not found a partner yet
    uint32_t pl[SS_PF], pr[SS_PF];           // tiles on their way (memory path only): pl[u] = [lc + 64 u, ..), pr[u] = [rc - 64 (u + 1), ..)
    if (!LDS) {
#pragma unroll
        for (int u = 0; u < SS_PF; ++u) {
            const int i = lo + 64 * u + lane, j = hi - 64 * (u + 1) + lane;
            pl[u] = i < hi ? c.A[i] : 0u;
            pr[u] = j >= lo ? c.A[j] : 0u;
        }
    }
    for (;;) {
        if (LQ == 0) {
            if (lc >= rc) break;
            const int n = min(64, rc - lc);
            validL = lane < n;
            baseL = lc;
            if (LDS) vL = validL ? c.sbuf[lc + lane - c.ldsFirst] : 0u;
            else {
                vL = pl[0];
#pragma unroll
                for (int u = 0; u + 1 < SS_PF; ++u) pl[u] = pl[u + 1];
                const int i = lc + 64 * SS_PF + lane;
                pl[SS_PF - 1] = i < hi ? c.A[i] : 0u;
            }
            lc += n;
            LQ = __ballot(validL && ssK(vL) >= Kp);
        }
        if (RQ == 0) {
            if (lc >= rc) break;
            const int n = min(64, rc - lc);
            validR = lane >= 64 - n;           // a short tile (the last one) fills the top lanes: lane l <-> position rc - 64 + l either way
            baseR = rc - 64;
            if (LDS) vR = validR ? c.sbuf[baseR + lane - c.ldsFirst] : 0u;
            else {
                vR = pr[0];
#pragma unroll
                for (int u = 0; u + 1 < SS_PF; ++u) pr[u] = pr[u + 1];
                const int j = rc - 64 * (SS_PF + 1) + lane;
                pr[SS_PF - 1] = j >= lo ? c.A[j] : 0u;
            }
            rc -= n;
            RQ = __ballot(validR && ssK(vR) <= Kp);
        }
        if (LQ != 0 && RQ != 0) {
            // pair the pending stoppers by rank: the k-th from the left with the k-th from the right -- all of them lie on their own side of
            // the unread part, so every pair swaps
            const bool isL = (LQ >> lane) & 1ull, isR = (RQ >> lane) & 1ull;
            const int nl = __popcll(LQ), nr = __popcll(RQ), k = min(nl, nr);
            const int rl = ss_rank_below(LQ);
            const int rr = nr - ss_rank_below(RQ) - (isR ? 1 : 0);
            const bool goL = isL && rl < k, goR = isR && rr < k;
            if (goL) c.xl[rl] = vL;
            if (goR) c.xr[rr] = vR;
            __builtin_amdgcn_wave_barrier();
            if (goL) { vL = c.xr[rl]; ss_st<LDS>(c, baseL + lane, vL); }
            if (goR) { vR = c.xl[rr]; ss_st<LDS>(c, baseR + lane, vR); }
            __builtin_amdgcn_wave_barrier();
            LQ = __ballot(isL && rl >= k);
            RQ = __ballot(isR && rr >= k);
        }
    }
    // everything has been read; at most one side still has stoppers, and they sit in that side's last tile
    if (LQ != 0) {
        // the right scan walks into the left side's last tile from above: the zone is [first pending stopper, end of that tile)
        const int l0 = (int)__builtin_ctzll(LQ);
        const unsigned long long LE = __ballot(validL && lane >= l0 && ssK(vL) <= Kp);
        return ss_zone<LDS>(c, vL, baseL, LQ, LE, lc - baseL);
    }
    if (RQ != 0) {
        // the left scan walks into the right side's last tile from below: the zone is [start of that tile, last pending stopper]
        const int r0 = 63 - (int)__builtin_clzll(RQ);
        const unsigned long long GE = __ballot(validR && lane <= r0 && ssK(vR) >= Kp);
        return ss_zone<LDS>(c, vR, baseR, GE, RQ, r0 + 1);
    }
    return lc;
}

// libstdc++ heap sort of [first, last) (std::__partial_sort(first, last, last) = __make_heap + __sort_heap), one lane
template <bool LDS>
__device__ void ss_adjust_heap(const SsCtx& c, int first, int hole, int len, uint32_t value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (ssK(ss_ld<LDS>(c, first + child)) < ssK(ss_ld<LDS>(c, first + child - 1))) --child;
        ss_st<LDS>(c, first + hole, ss_ld<LDS>(c, first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        ss_st<LDS>(c, first + hole, ss_ld<LDS>(c, first + child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && ssK(ss_ld<LDS>(c, first + parent)) < ssK(value)) {
        ss_st<LDS>(c, first + hole, ss_ld<LDS>(c, first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    ss_st<LDS>(c, first + hole, value);
}
template <bool LDS>
__device__ void ss_heapsort(const SsCtx& c, int first, int last)
{
    const int len = last - first;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; --parent) {
        ss_adjust_heap<LDS>(c, first, parent, len, ss_ld<LDS>(c, first + parent));
        if (parent == 0) break;
    }
    while (last - first > 1) {
        --last;
        const uint32_t value = ss_ld<LDS>(c, last);
        ss_st<LDS>(c, last, ss_ld<LDS>(c, first));
        ss_adjust_heap<LDS>(c, first, 0, last - first, value);
    }
}

// levels the loop spends on a range of m equal keys (its larger child has ceil((m - 1) / 2) + (m even ? 0 : 0) ... elements: see ss_emit_equal)
__device__ __forceinline__ int ss_equal_levels(int m)
{
    int lv = 0;
    while (m > 16) { const int left = 1 + (m - 1) / 2, right = m - left; m = max(left, right); ++lv; }
    return lv;
}

// A range [first, last) of equal keys.  One level of the loop on it: no comparison is ever true, so the median step swaps *first with the middle
// element, the partition swaps the j-th element of [first + 1, last) with the j-th from its end until they meet (a reversal), and the cut is
// first + 1 + (m - 1) / 2.  Position p therefore goes to: first -> first + last - mid, mid -> first, any other p -> first + last - p.  Every
// element follows its own position down to a leaf; the final insertion sort moves nothing (all keys equal).
template <bool LDS>
__device__ __forceinline__ int ss_emit_equal(const SsCtx& c, int first, int last, int outPos)
{
    const int m = last - first;
    for (int t0 = 0; t0 < m; t0 += 256) {
        uint32_t v[4];
        int p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { p[u] = first + t0 + 64 * u + c.lane; v[u] = p[u] < last ? ss_ld<LDS>(c, p[u]) : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p[u] < last) {
                int f = first, l = last, q = p[u];
                while (l - f > 16) {
                    const int mm = l - f, mid = f + mm / 2;
                    q = q == f ? f + l - mid : q == mid ? f : f + l - q;
                    const int cut = f + 1 + (mm - 1) / 2;
                    if (q < cut) l = cut; else f = cut;
                }
                c.out[outPos + q - first] = v[u];
            }
        }
    }
    return outPos + m;
}

// a leaf of the loop (<= 16 elements): __final_insertion_sort = stable sort by key; the elements that are listed (K <= Kthr) come first
template <bool LDS>
__device__ __forceinline__ int ss_emit_leaf(const SsCtx& c, int first, int last, uint32_t Kthr, int outPos)
{
    const int m = last - first, lane = c.lane;
    const uint32_t v = lane < m ? ss_ld<LDS>(c, first + lane) : 0xffffffffu;
    const uint32_t k = ssK(v);
    int rank = 0;
    for (int j = 0; j < m; ++j) {
        const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)k, j);
        rank += (kj < k || (kj == k && j < lane)) ? 1 : 0;
    }
    const bool listed = lane < m && k <= Kthr;
    if (listed) c.out[outPos + rank] = v;
    return outPos + (int)__popcll(__ballot(listed));
}

// a sorted range: its listed elements are a prefix
template <bool LDS>
__device__ __forceinline__ int ss_emit_sorted(const SsCtx& c, int first, int last, uint32_t Kthr, int outPos)
{
    int cnt = 0;
    for (int p = first + c.lane; p - c.lane < last; p += 64) {
        const uint32_t v = p < last ? ss_ld<LDS>(c, p) : 0xffffffffu;
        const bool listed = p < last && ssK(v) <= Kthr;
        if (listed) c.out[outPos + p - first] = v;
        cnt += (int)__popcll(__ballot(listed));
    }
    return outPos + cnt;
}

__global__ __launch_bounds__(64) void k_lsd_seedsort(const LineGeom* __restrict__ gp, uint32_t* keysInAll, uint32_t* keysOutAll, int* __restrict__ keyCount,
                                                     const int* __restrict__ maxN, int nOverride, int kthrOverride, int depthOverride)
{
    __shared__ uint32_t s_buf[SS_CAP];
    __shared__ uint32_t s_xl[64], s_xr[64];
    const LineGeom& g = *gp;
    const int img = blockIdx.x, lane = threadIdx.x;
    SsCtx c;
    c.A = keysInAll + (size_t)img * g.Ps;
    c.out = keysOutAll + (size_t)img * g.Ps;
    c.sbuf = s_buf; c.xl = s_xl; c.xr = s_xr; c.ldsFirst = 0; c.lane = lane;
    const int n = nOverride >= 0 ? nOverride : (g.Ws - 1) * (g.Hs - 1);
    uint32_t Kthr;
    if (kthrOverride >= 0) Kthr = (uint32_t)kthrOverride;
    else {
        const int mN = maxN[img * 32];
        if (mN <= 0) { if (lane == 0) keyCount[img * 32] = 0; return; }      // no defined pixel: no seed
        // the smallest bin a defined pixel can fall into (ll_angle: bin = int(norm * bin_coef), norm > rho <=> gx^2 + gy^2 >= nThr), as k_lsd_keys bins it
        const double max_grad = sqrt((double)mN / 4.0);
        const double bin_coef = (double)(g.nBins - 1) / max_grad;
        const double normT = sqrt((double)g.nThr / 4.0);
        const int binT = (int)(normT * bin_coef);
        Kthr = (uint32_t)(g.nBins - 1 - binT);
    }
    if (n <= 0) { if (lane == 0) keyCount[img * 32] = 0; return; }
    const int depth0 = depthOverride >= 0 ? depthOverride : 2 * (31 - __builtin_clz((unsigned)n));
    // the ranges still to do (right siblings on the path), one per lane: at most depth0 + 1 <= 43 of them
    int stF = 0, stL = 0, stD = 0;
    uint32_t stLb = 0, stUb = 0;
    int sp = 0;
#define SS_PUSH(F, L, D, LB, UB) do { if (lane == sp) { stF = (F); stL = (L); stD = (D); stLb = (LB); stUb = (UB); } ++sp; } while (0)
    SS_PUSH(0, n, depth0, 0u, (uint32_t)(g.nBins - 1));
    int outPos = 0;
    bool inLDS = false;
    int ldsLast = 0;
    while (sp > 0) {
        --sp;
        int first = __builtin_amdgcn_readlane(stF, sp), last = __builtin_amdgcn_readlane(stL, sp), depth = __builtin_amdgcn_readlane(stD, sp);
        uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)stLb, sp), ub = (uint32_t)__builtin_amdgcn_readlane((int)stUb, sp);     // lb <= K <= ub for every element of the range
        if (inLDS && first >= ldsLast) inLDS = false;
        for (;;) {
            const int m = last - first;
            if (lb > Kthr) break;                                     // only undefined pixels: never seeds, never leave the range
            if (m <= 16) { outPos = inLDS ? ss_emit_leaf<true>(c, first, last, Kthr, outPos) : ss_emit_leaf<false>(c, first, last, Kthr, outPos); break; }
            if (lb == ub && ss_equal_levels(m) <= depth) { outPos = inLDS ? ss_emit_equal<true>(c, first, last, outPos) : ss_emit_equal<false>(c, first, last, outPos); break; }
            if (depth == 0) {
                if (lane == 0) { if (inLDS) ss_heapsort<true>(c, first, last); else ss_heapsort<false>(c, first, last); }
                __builtin_amdgcn_wave_barrier();
                outPos = inLDS ? ss_emit_sorted<true>(c, first, last, Kthr, outPos) : ss_emit_sorted<false>(c, first, last, Kthr, outPos);
                break;
            }
            if (!inLDS && m <= SS_CAP) {
                for (int i0 = 0; i0 < m; i0 += 256) {
                    uint32_t t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = i0 + 64 * u + lane < m ? c.A[first + i0 + 64 * u + lane] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (i0 + 64 * u + lane < m) s_buf[i0 + 64 * u + lane] = t[u];
                }
                __builtin_amdgcn_wave_barrier();
                inLDS = true; c.ldsFirst = first; ldsLast = last;
            }
            --depth;
            // __move_median_to_first(first, first + 1, mid, last - 1)
            const int mid = first + m / 2;
            const int pidx = lane == 0 ? first : lane == 1 ? first + 1 : lane == 2 ? mid : last - 1;
            uint32_t pv = 0;
            if (lane < 4) pv = inLDS ? s_buf[pidx - c.ldsFirst] : c.A[pidx];
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)pv, 0), ea = (uint32_t)__builtin_amdgcn_readlane((int)pv, 1);
            const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)pv, 2), ec = (uint32_t)__builtin_amdgcn_readlane((int)pv, 3);
            const uint32_t Ka = ssK(ea), Kb = ssK(eb), Kc = ssK(ec);
            int sel;      // 0: a, 1: b, 2: c
            if (Ka < Kb) sel = Kb < Kc ? 1 : (Ka < Kc ? 2 : 0);
            else sel = Ka < Kc ? 0 : (Kb < Kc ? 2 : 1);
            const uint32_t es = sel == 0 ? ea : sel == 1 ? eb : ec;
            const int sidx = sel == 0 ? first + 1 : sel == 1 ? mid : last - 1;
            if (lane == 0) {
                if (inLDS) { s_buf[first - c.ldsFirst] = es; s_buf[sidx - c.ldsFirst] = e0; }
                else { c.A[first] = es; c.A[sidx] = e0; }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t Kp = ssK(es);
            const int cut = inLDS ? ss_partition<true>(c, first + 1, last, Kp) : ss_partition<false>(c, first + 1, last, Kp);
            SS_PUSH(cut, last, depth, max(lb, Kp), ub);               // [cut, last): K >= Kp
            last = cut; ub = min(ub, Kp);                             // [first, cut): K <= Kp (the pivot sits at first)
        }
    }
#undef SS_PUSH
    if (lane == 0) keyCount[img * 32] = outPos;
}

int launch_lsd_seedsort(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, int kthrOverride, int depthOverride)
{
    hipLaunchKernelGGL(k_lsd_seedsort, dim3(n_images), dim3(64), 0, s, b.geom, b.keysA, b.keysB, b.keyCount, b.maxN, nOverride, kthrOverride, depthOverride);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
