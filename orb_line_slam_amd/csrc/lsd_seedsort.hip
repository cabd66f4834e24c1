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
#pragma clang diagnostic ignored "-Winline-asm"      // ss_async_ld names m0 (the LDS base of global_load_lds) as clobbered

namespace olf {

// the grid-wide top levels (launch_seedsort_top, below)
#ifndef OLF_SS_TOP_LEVELS
#define OLF_SS_TOP_LEVELS 8
#endif
#ifndef OLF_SS_TOP_MIN
#define OLF_SS_TOP_MIN 8192
#endif
constexpr int SS_TOP_MIN = OLF_SS_TOP_MIN, SS_TOP_JOBS = 128, SS_TOP_LEVELS = OLF_SS_TOP_LEVELS, SS_TOP_FINAL = 512;
constexpr int SS_JW = 12;      // words of a job: first, last, depth of its children, lb, ub, pivot key, first tile, tiles, s, cut
constexpr int SS_TOP_WORDS = 8 + 2 * SS_TOP_JOBS * SS_JW + SS_TOP_FINAL * 5;      // per image: counters [nJobs, nNext, nFinal, tiles, Kthr, n], two job lists, final entries
constexpr int SS_CAP = 1024;      // elements of a range held in LDS (4 KB + 2 KB of exchange arrays: 24 waves = 24 images per CU)
constexpr int SS_NE_MEM = 4;      // tiles per block while a range streams from memory (stopper queues: 2 x 256 pairs = the idle range buffer; the staged blocks behind it)
constexpr int SS_NE_LDS = 2;      // ... while it is LDS resident (stopper queues: 2 x 128 pairs behind the range buffer)

struct SsCtx {
    uint32_t* A;          // the image's keys in memory; sorted in place
    uint32_t* sbuf;       // LDS copy of [ldsFirst, ldsLast)
    uint32_t* out;        // the seed list
    int ldsFirst;
    int lane;
};

__device__ __forceinline__ uint32_t ssK(uint32_t e) { return e >> 22; }
// wave-uniform values are pinned to scalar registers: the divergence analysis otherwise gives up on the block counters (they flow through
// loops with per-lane conditions), and every branch on them becomes an exec-mask sequence with vector copies of the loop state
__device__ __forceinline__ int ssU(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <bool LDS> __device__ __forceinline__ uint32_t ss_ld(const SsCtx& c, int i) { return LDS ? c.sbuf[i - c.ldsFirst] : c.A[i]; }
template <bool LDS> __device__ __forceinline__ void ss_st(const SsCtx& c, int i, uint32_t v) { if (LDS) c.sbuf[i - c.ldsFirst] = v; else c.A[i] = v; }
// one tile of the next block on its way from memory into LDS (global_load_lds_dword: lane l's word lands at lds_byte_off + 4 l).  Inline
// assembly on purpose: issued through the builtin, the compiler's wait-count pass puts an s_waitcnt vmcnt(0) in front of the next LDS access
// of any kind (it cannot tell the queues from the staging area), which turns the prefetch into a blocking load.  The data is only read after
// an explicit s_waitcnt vmcnt(0); the compiler's own waits stay safe because vector memory operations complete in order.
__device__ __forceinline__ void ss_async_ld(const uint32_t* g, unsigned lds_byte_off)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(lds_byte_off) : "memory", "m0");
}
// NE tiles of one block whose positions are all inside the range: ONE address and one LDS base, the tiles through the instruction's offset field (it advances
// the memory address and the LDS address together) -- 2 + NE instructions where NE single loads take 6 NE (clamp, 64-bit address, m0, wait state, load)
template <int NE>
__device__ __forceinline__ void ss_async_ld_block(const uint32_t* g, unsigned lds_byte_off)
{
    static_assert(NE == 1 || NE == 2 || NE == 4 || NE == 8, "tiles per block");
    if (NE == 1)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(lds_byte_off) : "memory", "m0");
    else if (NE == 2)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\tglobal_load_lds_dword %0, off offset:256" :: "v"(g), "s"(lds_byte_off) : "memory", "m0");
    else if (NE == 4)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\tglobal_load_lds_dword %0, off offset:256\n\tglobal_load_lds_dword %0, off offset:512\n\t"
                     "global_load_lds_dword %0, off offset:768" :: "v"(g), "s"(lds_byte_off) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\tglobal_load_lds_dword %0, off offset:256\n\tglobal_load_lds_dword %0, off offset:512\n\t"
                     "global_load_lds_dword %0, off offset:768\n\tglobal_load_lds_dword %0, off offset:1024\n\tglobal_load_lds_dword %0, off offset:1280\n\t"
                     "global_load_lds_dword %0, off offset:1536\n\tglobal_load_lds_dword %0, off offset:1792" :: "v"(g), "s"(lds_byte_off) : "memory", "m0");
}
__device__ __forceinline__ unsigned ss_lds_off(const void* p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(p)); }
__device__ __forceinline__ int ss_rank_below(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// ---- the Hoare partition, NE x 64 elements per side and step -----------------------------------------------------------------------------
// A block is NE tiles held in registers, slot u / lane l <-> position base + 64 u + l.  Its stoppers are ranked in scan order once when
// the block is read (the left scan's: elements >= pivot by ascending position; the right scan's: elements <= pivot by descending position);
// c? = stoppers of the block that have found their partner.  A step pairs min(nL - cL, nR - cR) pending stoppers by rank through the LDS
// exchange arrays and stores the swapped values; the side that runs out of stoppers reads its next block.  Many independent element chains
// per step is what a single wave needs: the 64-element version of this loop spent 870 cycles per tile on dependent latency.
template <int NE> struct SsBlock {
    uint32_t v[NE];
    int rk[NE];           // rank of the element among the block's stoppers, -1: not a stopper
    int base, n, c;       // position of slot 0 / lane 0; stoppers; stoppers consumed
    int lim;              // left block: positions < lim are valid; right block: positions >= lim
};

// The meeting zone of the two scans lies inside the last block of the side that still has stoppers.  G: the zone's elements >= pivot with
// rank jG from the left (jG < 0: not in G), LE: its elements <= pivot with rank jLE from the right.  Pair j swaps while L[j] < R[j]; the
// partition's return value is min(L[s], R[s - 1]) after s swaps, or zoneEnd if neither exists.
template <bool LDS, int NE>
__device__ __forceinline__ int ss_zone(const SsCtx& c, uint32_t* XL, uint32_t* XR, SsBlock<NE>& B, const int* jG, int nG, const int* jLE, int nLE, int zoneEnd)
{
    const int lane = c.lane;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int pos = B.base + 64 * u + lane;
        if (jG[u] >= 0) XL[jG[u]] = (uint32_t)pos;
        if (jLE[u] >= 0) XR[jLE[u]] = (uint32_t)pos;
    }
    __builtin_amdgcn_wave_barrier();
    bool swG[NE], swLE[NE];
    int s = 0;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int pos = B.base + 64 * u + lane;
        swG[u] = false; swLE[u] = false;
        if (jG[u] >= 0 && jG[u] < nLE) swG[u] = pos < (int)XR[jG[u]];
        if (jLE[u] >= 0 && jLE[u] < nG) swLE[u] = (int)XL[jLE[u]] < pos;       // (an element equal to the pivot is in both sets but can only swap as one of them)
        s += (int)__popcll(wave_vote(swG[u]));
    }
    s = ssU(s);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        if (swG[u]) XL[jG[u]] = B.v[u];
        if (swLE[u]) XR[jLE[u]] = B.v[u];
    }
    __builtin_amdgcn_wave_barrier();
    int cut = zoneEnd;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int pos = B.base + 64 * u + lane;
        if (swG[u]) { B.v[u] = XR[jG[u]]; ss_st<LDS>(c, pos, B.v[u]); }
        if (swLE[u]) { B.v[u] = XL[jLE[u]]; ss_st<LDS>(c, pos, B.v[u]); }
        const unsigned long long cand = wave_vote(jG[u] == s || (s >= 1 && jLE[u] == s - 1));
        if (cand) cut = min(cut, B.base + 64 * u + (int)__builtin_ctzll(cand));
    }
    __builtin_amdgcn_wave_barrier();
    return ssU(cut);
}

// std::__unguarded_partition(lo, hi, pivot) on [lo, hi) (= [first + 1, last)) with comp(a, b) = K(a) < K(b); returns the cut.
// When a block is read, its stoppers go to the side's queue in LDS as (position, value) pairs in scan order (QL / QR, 64 NE entries each); a
// step swaps the first k = min(pending left, pending right) pairs with consecutive lanes -- dense over the stoppers, no per-slot bookkeeping --
// and the side whose queue is empty reads its next block.  dropRight: the part right of the cut will never be looked at again (it can only hold
// undefined pixels), so what would be swapped into it is not stored.
// LDS: the range lives in c.sbuf; otherwise it streams from memory, the next block of either side already on its way into SL / SR.
template <bool LDS, int NE>
__device__ __forceinline__ int ss_partition(const SsCtx& c, uint2* QL, uint2* QR, uint32_t* SL, uint32_t* SR, int lo, int hi, uint32_t Kp, bool dropRight)
{
    const int lane = c.lane;
    lo = ssU(lo); hi = ssU(hi); Kp = (uint32_t)ssU((int)Kp);
    int lc = lo, rc = hi;                    // unread: [lc, rc)
    // memory path: the block after the current one is on its way into LDS (SL / SR, global_load_lds: no register, no compiler-placed wait --
    // loads kept in registers ended up behind an s_waitcnt vmcnt(0) right after their issue).  pfL / pfR: the position it starts at, -1: none
    int pfL = -1, pfR = -1;
    SsBlock<NE> L, R;
    L.n = L.c = R.n = R.c = 0; L.base = lo; R.base = hi; L.lim = lo; R.lim = hi;
#pragma unroll
    for (int u = 0; u < NE; ++u) { L.v[u] = R.v[u] = 0u; L.rk[u] = R.rk[u] = -1; }
    for (;;) {
        if (L.c == L.n) {
            if (lc >= rc) break;
            const int n = min(64 * NE, rc - lc);
            L.base = lc; L.lim = lc + n;
            if (!LDS && pfL == lc) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < NE; ++u) L.v[u] = SL[64 * u + lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the staged block is in registers before the next one may overwrite it
            } else {
                // (positions past the block are read clamped: no per-slot branch; their values are never looked at)
#pragma unroll
                for (int u = 0; u < NE; ++u) L.v[u] = ss_ld<LDS>(c, min(lc + 64 * u + lane, L.lim - 1));
            }
            if (!LDS) {
                pfL = -1;
                if (lc + n < rc) {
                    pfL = lc + n;
#pragma unroll
                    for (int u = 0; u < NE; ++u) { if (pfL + 64 * NE <= hi) { if (u == 0) ss_async_ld_block<NE>(c.A + pfL + lane, ss_lds_off(SL)); }
                                                   else ss_async_ld(c.A + min(pfL + 64 * u + lane, hi - 1), ss_lds_off(SL) + 256u * u); }
                }
            }
            int run = 0;
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int pos = lc + 64 * u + lane;
                // (votes per comparison, combined as lane masks: a vote on the combined predicate materialises the bool -- a v_cndmask + v_cmp pair per tile)
                const unsigned long long m = wave_vote(pos < L.lim) & wave_vote(ssK(L.v[u]) >= Kp);
                const bool st = wave_bit(m);
                L.rk[u] = st ? run + ss_rank_below(m) : -1;
                if (st) QL[L.rk[u]] = make_uint2((uint32_t)pos, L.v[u]);
                run += (int)__popcll(m);
            }
            L.n = ssU(run); L.c = 0;
            lc += n;
        }
        if (R.c == R.n) {
            if (lc >= rc) break;
            const int n = min(64 * NE, rc - lc);
            R.base = rc - 64 * NE; R.lim = rc - n;      // a short block (the last one) fills the top slots
            if (!LDS && pfR == R.base) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < NE; ++u) R.v[u] = SR[64 * u + lane];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int u = 0; u < NE; ++u) R.v[u] = ss_ld<LDS>(c, max(R.base + 64 * u + lane, R.lim));
            }
            if (!LDS) {
                pfR = -1;
                if (lc < rc - n) {
                    pfR = rc - n - 64 * NE;
#pragma unroll
                    for (int u = 0; u < NE; ++u) { if (pfR >= lo) { if (u == 0) ss_async_ld_block<NE>(c.A + pfR + lane, ss_lds_off(SR)); }
                                                   else ss_async_ld(c.A + max(pfR + 64 * u + lane, lo), ss_lds_off(SR) + 256u * u); }
                }
            }
            int run = 0;
#pragma unroll
            for (int u = NE - 1; u >= 0; --u) {
                const int pos = R.base + 64 * u + lane;
                const unsigned long long m = wave_vote(pos >= R.lim) & wave_vote(ssK(R.v[u]) <= Kp);
                const bool st = wave_bit(m);
                const int cnt = (int)__popcll(m);
                R.rk[u] = st ? run + cnt - 1 - ss_rank_below(m) : -1;
                if (st) QR[R.rk[u]] = make_uint2((uint32_t)pos, R.v[u]);
                run += cnt;
            }
            R.n = ssU(run); R.c = 0;
            rc -= n;
        }
        if (L.c < L.n && R.c < R.n) {
            // all pending stoppers lie on their own side of the unread part, so every pair swaps
            const int k = ssU(min(L.n - L.c, R.n - R.c));
            __builtin_amdgcn_wave_barrier();
            for (int j = lane; j < k; j += 64) {
                const uint2 a = QL[L.c + j], b = QR[R.c + j];
                ss_st<LDS>(c, (int)a.x, b.y);
                if (!dropRight) ss_st<LDS>(c, (int)b.x, a.y);
            }
            __builtin_amdgcn_wave_barrier();
            L.c += k; R.c += k;
        }
    }
    if (!LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no block may still be on its way into LDS when the buffers are reused
    // everything has been read; at most one side still has stoppers, and they sit in that side's last block (whose pending elements have not
    // been touched by the swaps above, so the register copy of the block is current where it matters)
    uint32_t* XL = reinterpret_cast<uint32_t*>(QL);
    uint32_t* XR = reinterpret_cast<uint32_t*>(QR);
    if (L.c < L.n) {
        // the right scan walks into the left side's last block from above: the zone is [first pending stopper, end of that block)
        int l0 = lc;
#pragma unroll
        for (int u = 0; u < NE; ++u) { const unsigned long long mm = wave_vote(L.rk[u] == L.c); if (mm) l0 = L.base + 64 * u + (int)__builtin_ctzll(mm); }
        int jG[NE], jLE[NE], run = 0;
#pragma unroll
        for (int u = NE - 1; u >= 0; --u) {
            const int pos = L.base + 64 * u + lane;
            const bool le = pos < L.lim && pos >= l0 && ssK(L.v[u]) <= Kp;
            const unsigned long long m = wave_vote(le);
            const int cnt = (int)__popcll(m);
            jLE[u] = le ? run + cnt - 1 - ss_rank_below(m) : -1;
            run += cnt;
            jG[u] = L.rk[u] >= L.c ? L.rk[u] - L.c : -1;
        }
        return ss_zone<LDS, NE>(c, XL, XR, L, jG, L.n - L.c, jLE, ssU(run), lc);
    }
    if (R.c < R.n) {
        // the left scan walks into the right side's last block from below: the zone is [start of that block, last pending stopper]
        int r0 = rc;
#pragma unroll
        for (int u = 0; u < NE; ++u) { const unsigned long long mm = wave_vote(R.rk[u] == R.c); if (mm) r0 = R.base + 64 * u + (int)__builtin_ctzll(mm); }
        int jG[NE], jLE[NE], run = 0;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int pos = R.base + 64 * u + lane;
            const bool ge = pos >= R.lim && pos <= r0 && ssK(R.v[u]) >= Kp;
            const unsigned long long m = wave_vote(ge);
            jG[u] = ge ? run + ss_rank_below(m) : -1;
            run += (int)__popcll(m);
            jLE[u] = R.rk[u] >= R.c ? R.rk[u] - R.c : -1;
        }
        return ss_zone<LDS, NE>(c, XL, XR, R, jG, ssU(run), jLE, R.n - R.c, r0 + 1);
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
    return outPos + (int)__popcll(wave_vote(listed));
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
        cnt += (int)__popcll(wave_vote(listed));
    }
    return outPos + cnt;
}

// -DOLF_SS_PROF: cycles and counts per phase of image 0 into status[16..] (tools/prof_seedsort.py)
#ifdef OLF_SS_PROF
enum { SP_PART_MEM = 0, SP_PART_LDS, SP_EQUAL, SP_LEAF, SP_LOAD, SP_PIVOT, SP_OTHER, SP_N_PART_MEM, SP_N_PART_LDS, SP_N_EQUAL, SP_N_LEAF, SP_N_LOAD, SP_V_PART_MEM, SP_V_PART_LDS, SP_V_EQUAL, SP_N };
#define SSPROF(i) do { const long long _t = __builtin_readcyclecounter(); sp_acc[i] += _t - sp_t; sp_t = _t; } while (0)
#define SSCNT(i, v) (sp_acc[i] += (v))
#else
#define SSPROF(i)
#define SSCNT(i, v)
#endif

// The sort of one image.  NW = 1: one wave, the ranges still to do on a stack held one per lane in registers (the batch kernel).  NW > 1: the NW waves
// of a workgroup share the image -- after a partition the two parts are independent, and a finished element's place in the seed list is its place in
// the sorted array, so no wave needs to know what the others have emitted: ranges that stream from memory go through a stack in LDS that any idle
// wave pops from, a range that fits a wave's LDS buffer is finished by that wave alone (its own register stack).  NEM: tiles per streamed block --
// one wave alone is bound by the latency of a block step, not by issue slots, so the few-images variant uses larger blocks.
// NI > 1 (= NW): the workgroup sorts NI images, one per wave to begin with, and the shared stack holds the streamed ranges of all of them (the image's
// slot travels in the entry's depth word): a wave that runs out of work takes over a range of a neighbour's image, so the launch ends near the *mean*
// sorting time of the images of a CU instead of the slowest one's (the big batch).
template <int NW, int NEM, int NI = 1>
__device__ __forceinline__ void ss_sort_image(const LineGeom& g, int img, int n_images, uint32_t* keysInAll, uint32_t* keysOutAll, int* __restrict__ keyCount,
                                              const int* __restrict__ maxN, int* __restrict__ status, int nOverride, int kthrOverride, int depthOverride,
                                              uint32_t* s_buf, uint32_t* s_x, int* ctl, int* topImg = nullptr, int grp = 0, int Gs = 1)
{
    // (grp, Gs: NI == 1 behind the grid-wide top levels only -- Gs workgroups share one image: a wave that finds its workgroup's stack empty takes the next of the
    // ranges the top levels left from the image's list (a counter in global memory); the ranges are disjoint in the key array and in the seed list, so nothing but
    // that counter and the seed count is shared, and the result does not depend on who sorts which range)
    static_assert(NI == 1 || NI == NW, "one wave per image of the group");
#ifdef OLF_SS_PROF
    long long sp_acc[SP_N] = {0}, sp_t = __builtin_readcyclecounter();
#endif
    constexpr int CAP = NEM > SS_NE_MEM ? 4 * 64 * NEM : SS_CAP;          // elements of a range a wave keeps in LDS (= the words of its range buffer)
    constexpr int SHCAP = NI > 1 ? 64 : 256;                              // entries of the shared stack (NW > 1)
    const int lane = threadIdx.x & 63;
    const int img0 = img;                                                 // NI > 1: the group's first image; this wave starts on image img0 + wave
    if (NI > 1) img = img0 + ssU((int)(threadIdx.x >> 6));      // (pinned: the wave index is uniform, the compiler does not know it)
    SsCtx c;
    c.A = keysInAll + (size_t)img * g.Ps;
    c.out = keysOutAll + (size_t)img * g.Ps;
    c.sbuf = s_buf; c.ldsFirst = 0; c.lane = lane;
    const int n = nOverride >= 0 ? nOverride : (g.Ws - 1) * (g.Hs - 1);
    uint32_t Kthr = 0;
    bool empty = n <= 0 || img >= n_images;
    if (empty) {}
    else if (kthrOverride >= 0) Kthr = (uint32_t)kthrOverride;
    else {
        const int mN = maxN[img * 32];
        if (mN <= 0) empty = true;                                        // no defined pixel: no seed
        else {
            // the smallest bin a defined pixel can fall into (ll_angle: bin = int(norm * bin_coef), norm > rho <=> gx^2 + gy^2 >= nThr), as k_lsd_keys bins it
            const double max_grad = sqrt((double)mN / 4.0);
            const double bin_coef = (double)(g.nBins - 1) / max_grad;
            const double normT = sqrt((double)g.nThr / 4.0);
            const int binT = (int)(normT * bin_coef);
            Kthr = (uint32_t)(g.nBins - 1 - binT);
        }
    }
    if (NI == 1 && empty) { if (threadIdx.x == 0 && grp == 0) keyCount[img * 32] = 0; return; }
    if (NI == 1 && Gs > 1 && !topImg && grp != 0) return;      // (no ranges to share: the first workgroup sorts the image)
    const int depth0 = depthOverride >= 0 ? depthOverride : 2 * (31 - __builtin_clz((unsigned)n));
    // the ranges still to do (right siblings on the path), one per lane: at most depth0 + 1 <= 43 of them
    int stF = 0, stL = 0, stD = 0;
    uint32_t stLb = 0, stUb = 0;
    int sp = 0;
#define SS_PUSH(F, L, D, LB, UB) do { if (lane == sp) { stF = (F); stL = (L); stD = (D); stLb = (LB); stUb = (UB); } ++sp; } while (0)
    // shared stack (NW > 1): ctl[0] lock, [1] entries, [2] waves holding work, [3] seeds listed; then five arrays of SHCAP words
    int* const shF = ctl + 4; int* const shL = shF + SHCAP; int* const shD = shL + SHCAP; int* const shLb = shD + SHCAP; int* const shUb = shLb + SHCAP;
#define SS_LOCK() do { if (lane == 0) { int _sp = 0; while (atomicCAS(&ctl[0], 0, 1) != 0) { __builtin_amdgcn_s_sleep(2); if (++_sp > (1 << 22)) { atomicOr(status, 128); break; } } } __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#define SS_UNLOCK() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (lane == 0) atomicExch(&ctl[0], 0); } while (0)
    // NI > 1: per image slot the seeds listed so far and its Kthr, behind the stack's arrays
    int* const cntS = shUb + SHCAP; int* const kthS = cntS + 8; int* const okS = kthS + 8;
    int curSlot = NI > 1 ? ssU((int)(threadIdx.x >> 6)) : 0;
    if (NW == 1) SS_PUSH(0, n, depth0, 0u, (uint32_t)(g.nBins - 1));
    else if (NI == 1) {
        if (threadIdx.x == 0) {
            ctl[0] = 0; ctl[2] = 0; ctl[3] = 0;
            if (topImg) {
                // the top levels have been partitioned by the grid-wide kernels (launch_seedsort_top): start from the ranges they left
                const int nf = min(topImg[2], SHCAP);
                const int* f = topImg + 8 + 2 * SS_TOP_JOBS * SS_JW;
                ctl[1] = 0;              // (the stack starts empty: the waves pull the ranges the top levels left one by one ...)
                cntS[0] = 0;             // ... until the image's list is exhausted
            } else { ctl[1] = 1; shF[0] = 0; shL[0] = n; shD[0] = depth0; shLb[0] = 0; shUb[0] = g.nBins - 1; }
        }
        __syncthreads();
    } else {
        if (lane == 0) { cntS[curSlot] = 0; kthS[curSlot] = (int)Kthr; okS[curSlot] = empty ? 0 : 1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int e = 0;
            for (int i = NI - 1; i >= 0; --i)      // (slot 0 on top)
                if (okS[i]) { shF[e] = 0; shL[e] = n; shD[e] = depth0 | (i << 8); shLb[e] = 0; shUb[e] = g.nBins - 1; ++e; }
            ctl[0] = 0; ctl[1] = e; ctl[2] = 0; ctl[3] = 0;
        }
        __syncthreads();
    }
    int listedEnd = 0;                 // one past the last seed this wave has listed (of the image it is working on)
    bool inLDS = false, holding = false;
    int ldsLast = 0;
    int guard = 0;
    for (;;) {
        if (NW > 1 && ++guard > (1 << 24)) { if (lane == 0) atomicOr(status, 64); break; }      // watchdog: a wave that found neither work nor the end
        int first, last, depth;
        uint32_t lb, ub;
        if (sp > 0) {
            --sp;
            first = __builtin_amdgcn_readlane(stF, sp); last = __builtin_amdgcn_readlane(stL, sp); depth = __builtin_amdgcn_readlane(stD, sp);
            lb = (uint32_t)__builtin_amdgcn_readlane((int)stLb, sp); ub = (uint32_t)__builtin_amdgcn_readlane((int)stUb, sp);     // lb <= K <= ub for every element of the range
        } else if (NW == 1) break;
        else {
            inLDS = false;
            if (holding) { if (lane == 0) atomicSub(&ctl[2], 1); holding = false; }
            // An idle wave looks at the counters WITHOUT the lock and only takes it when there is an entry to pop or the end to confirm: idle waves
            // that poll through the lock form a convoy that a working wave with a range to push never gets into (seen with 3 idle waves of 4:
            // 4 M failed attempts in a row -- the hardware's timing is deterministic enough to phase-lock).
            for (int idle = 0;; ++idle) {
                const int e = ssU(__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const int bz = ssU(__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (e > 0 || bz == 0) break;
                if (NI == 1 && topImg && ssU(__hip_atomic_load(&cntS[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) break;      // (ranges left in the image's list)
                __builtin_amdgcn_s_sleep(16);
                if (idle > (1 << 21)) { if (lane == 0) atomicOr(status, 64); break; }
            }
            int got = 0, done = 0, f = 0, l = 0, d = 0, a = 0, b = 0;
            SS_LOCK();
            if (lane == 0) {
                const int e = atomicAdd(&ctl[1], 0);
                if (e > 0) { f = shF[e - 1]; l = shL[e - 1]; d = shD[e - 1]; a = shLb[e - 1]; b = shUb[e - 1]; atomicAdd(&ctl[2], 1); atomicExch(&ctl[1], e - 1); got = 1; }
                else if (NI == 1 && topImg && cntS[0] == 0) {
                    // the next range of the image's list (cnt[3], zero when the last top level ends)
                    const int q = atomicAdd(topImg + 3, 1);
                    if (q < topImg[2]) {
                        const int* fq = topImg + 8 + 2 * SS_TOP_JOBS * SS_JW + 5 * q;
                        f = fq[0]; l = fq[1]; d = fq[2]; a = fq[3]; b = fq[4]; atomicAdd(&ctl[2], 1); got = 1;
                    } else cntS[0] = 1;
                }
                else if (atomicAdd(&ctl[2], 0) == 0) done = 1;
            }
            SS_UNLOCK();
            got = ssU(got); done = ssU(done);
            holding = got != 0;
            if (done) break;
            if (!got) continue;
            first = ssU(f); last = ssU(l); depth = ssU(d); lb = (uint32_t)ssU(a); ub = (uint32_t)ssU(b);
            if (NI > 1) {
                const int slot = depth >> 8;
                depth &= 255;
                if (slot != curSlot) {
                    if (lane == 0 && listedEnd > 0) atomicMax(&cntS[curSlot], listedEnd);
                    listedEnd = 0; curSlot = slot;
                    c.A = keysInAll + (size_t)(img0 + slot) * g.Ps;
                    c.out = keysOutAll + (size_t)(img0 + slot) * g.Ps;
                }
                Kthr = (uint32_t)ssU(kthS[slot]);
            }
        }
        first = ssU(first); last = ssU(last); depth = ssU(depth); lb = (uint32_t)ssU((int)lb); ub = (uint32_t)ssU((int)ub);
        if (inLDS && first >= ldsLast) inLDS = false;
        for (;;) {
            const int m = last - first;
            if (lb > Kthr) break;                                     // only undefined pixels: never seeds, never leave the range
            SSPROF(SP_OTHER);
            // (a finished element's place in the seed list is its place in the sorted array: the listed elements are the array's prefix)
            int end = first;
            if (m <= 16) { end = inLDS ? ss_emit_leaf<true>(c, first, last, Kthr, first) : ss_emit_leaf<false>(c, first, last, Kthr, first); SSPROF(SP_LEAF); SSCNT(SP_N_LEAF, 1); }
            else if (lb == ub && ss_equal_levels(m) <= depth) { end = inLDS ? ss_emit_equal<true>(c, first, last, first) : ss_emit_equal<false>(c, first, last, first); SSPROF(SP_EQUAL); SSCNT(SP_N_EQUAL, 1); SSCNT(SP_V_EQUAL, m); }
            else if (depth == 0) {
                if (lane == 0) { if (inLDS) ss_heapsort<true>(c, first, last); else ss_heapsort<false>(c, first, last); }
                __builtin_amdgcn_wave_barrier();
                end = inLDS ? ss_emit_sorted<true>(c, first, last, Kthr, first) : ss_emit_sorted<false>(c, first, last, Kthr, first);
            } else end = -1;
            if (end >= 0) { end = ssU(end); if (end > first) listedEnd = max(listedEnd, end); break; }
            if (!inLDS && m <= CAP) {
                for (int i0 = 0; i0 < m; i0 += 256) {
                    uint32_t t[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t[u] = i0 + 64 * u + lane < m ? c.A[first + i0 + 64 * u + lane] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (i0 + 64 * u + lane < m) s_buf[i0 + 64 * u + lane] = t[u];
                }
                __builtin_amdgcn_wave_barrier();
                inLDS = true; c.ldsFirst = first; ldsLast = last;
                SSPROF(SP_LOAD); SSCNT(SP_N_LOAD, 1);
            }
            --depth;
            // __move_median_to_first(first, first + 1, mid, last - 1)
            const int mid = first + m / 2;
            const int pidx = lane == 0 ? first : lane == 1 ? first + 1 : lane == 2 ? mid : last - 1;
            uint32_t pv = 0;
            // (two branches, not one conditional expression: that becomes a generic-pointer flat load, slower than ds_read for the many LDS-resident ranges)
            if (inLDS) { if (lane < 4) pv = s_buf[pidx - c.ldsFirst]; asm volatile("" : "+v"(pv)); }
            else if (lane < 4) pv = c.A[pidx];
            const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)pv, 0), ea = (uint32_t)__builtin_amdgcn_readlane((int)pv, 1);
            const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)pv, 2), ec = (uint32_t)__builtin_amdgcn_readlane((int)pv, 3);
            const uint32_t Ka = ssK(ea), Kb = ssK(eb), Kc = ssK(ec);
            int sel;      // 0: a, 1: b, 2: c
            if (Ka < Kb) sel = Kb < Kc ? 1 : (Ka < Kc ? 2 : 0);
            else sel = Ka < Kc ? 0 : (Kb < Kc ? 2 : 1);
            const uint32_t es = sel == 0 ? ea : sel == 1 ? eb : ec;
            const int sidx = sel == 0 ? first + 1 : sel == 1 ? mid : last - 1;
            if (lane == 0) {
                if (inLDS) { s_buf[first - c.ldsFirst] = es; s_buf[sidx - c.ldsFirst] = e0; asm volatile("" ::: "memory"); }      // (no tail merging into a flat store)
                else { c.A[first] = es; c.A[sidx] = e0; asm volatile("" ::: "memory"); }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t Kp = ssK(es);
            SSPROF(SP_PIVOT);
            int cut;
            const bool dropRight = Kp > Kthr;
            uint2* const qx = reinterpret_cast<uint2*>(s_x);
            uint2* const qb = reinterpret_cast<uint2*>(s_buf);
            if (!inLDS) cut = ss_partition<false, NEM>(c, qb, qb + 64 * NEM, s_x, s_x + 64 * NEM, first + 1, last, Kp, dropRight);
            else if (m <= 65) cut = ss_partition<true, 1>(c, qx, qx + 64, nullptr, nullptr, first + 1, last, Kp, dropRight);       // small ranges: no work on empty slots
            else cut = ss_partition<true, SS_NE_LDS>(c, qx, qx + 64 * SS_NE_LDS, nullptr, nullptr, first + 1, last, Kp, dropRight);
            if (inLDS) { SSPROF(SP_PART_LDS); SSCNT(SP_N_PART_LDS, 1); SSCNT(SP_V_PART_LDS, m); } else { SSPROF(SP_PART_MEM); SSCNT(SP_N_PART_MEM, 1); SSCNT(SP_V_PART_MEM, m); }
            // [cut, last): K >= Kp -- for the wave itself when the range sits in its LDS buffer, else for whichever wave is idle
            if (NW == 1 || inLDS) SS_PUSH(cut, last, depth, max(lb, Kp), ub);
            else if (max(lb, Kp) <= Kthr) {
                int full = 0;
                SS_LOCK();
                if (lane == 0) {
                    const int e = atomicAdd(&ctl[1], 0);
                    if (e < SHCAP) { shF[e] = cut; shL[e] = last; shD[e] = NI > 1 ? depth | (curSlot << 8) : depth; shLb[e] = (int)max(lb, Kp); shUb[e] = (int)ub; atomicExch(&ctl[1], e + 1); }
                    else full = 1;
                }
                SS_UNLOCK();
                // a full stack: the wave keeps the range for itself (its own stack is emptied before it looks at the shared one again, so the
                // range is still of the image the wave is on)
                if (ssU(full)) SS_PUSH(cut, last, depth, max(lb, Kp), ub);
            }
            last = cut; ub = min(ub, Kp);                             // [first, cut): K <= Kp (the pivot sits at first)
        }
    }
#undef SS_PUSH
#undef SS_LOCK
#undef SS_UNLOCK
#ifdef OLF_SS_PROF
    SSPROF(SP_OTHER);
    if (NW == 1 && lane == 0 && img == 0) { long long* o = reinterpret_cast<long long*>(status + 16); for (int q = 0; q < SP_N; ++q) o[q] = sp_acc[q]; }
#endif
    if (NW == 1) { if (lane == 0) keyCount[img * 32] = listedEnd; }
    else if (NI > 1) {
        if (lane == 0 && listedEnd > 0) atomicMax(&cntS[curSlot], listedEnd);
        __syncthreads();
        if (threadIdx.x < NI && img0 + (int)threadIdx.x < n_images) keyCount[(img0 + (int)threadIdx.x) * 32] = cntS[threadIdx.x];
    } else {
        if (lane == 0) atomicMax(&ctl[3], listedEnd);
        __syncthreads();
        if (threadIdx.x == 0) { if (Gs > 1 && topImg) atomicMax(&keyCount[img * 32], ctl[3]); else keyCount[img * 32] = ctl[3]; }
    }
}

__global__ __launch_bounds__(64) void k_lsd_seedsort(const LineGeom* __restrict__ gp, uint32_t* keysInAll, uint32_t* keysOutAll, int* __restrict__ keyCount,
                                                     const int* __restrict__ maxN, int* __restrict__ status, int nOverride, int kthrOverride, int depthOverride)
{
    __shared__ __align__(8) uint32_t s_buf[SS_CAP];
    __shared__ __align__(8) uint32_t s_x[4 * 64 * SS_NE_LDS];      // LDS path: the two stopper queues; memory path: the two staged blocks
    static_assert(4 * 64 * SS_NE_MEM <= SS_CAP && 2 * 64 * SS_NE_MEM <= 4 * 64 * SS_NE_LDS, "memory path: queues in the range buffer, staged blocks in s_x");
    ss_sort_image<1, SS_NE_MEM>(*gp, blockIdx.x, gridDim.x, keysInAll, keysOutAll, keyCount, maxN, status, nOverride, kthrOverride, depthOverride, s_buf, s_x, nullptr);
}

// few images (the drop-in's online shape: one stereo pair per call): NW waves per image, blocks of NEM tiles
template <int NW, int NEM, int NI>
__global__ __launch_bounds__(64 * NW) void k_lsd_seedsort_mw(const LineGeom* __restrict__ gp, uint32_t* keysInAll, uint32_t* keysOutAll, int* __restrict__ keyCount,
                                                            const int* __restrict__ maxN, int* __restrict__ status, int nOverride, int kthrOverride, int depthOverride,
                                                            int n_images, int* __restrict__ topAll, int Gs)
{
    extern __shared__ __align__(8) uint32_t s_dyn[];
    constexpr int BUFW = 4 * 64 * NEM, XW = 2 * 64 * NEM;
    const int blk = NI == 1 ? (int)blockIdx.x / Gs : (int)blockIdx.x, grp = NI == 1 ? (int)blockIdx.x % Gs : 0;
    static_assert(XW >= 4 * 64 * SS_NE_LDS, "the LDS path's queues fit the staging area");
    const int wv = threadIdx.x >> 6;
    // the staging areas first: global_load_lds takes its LDS base from 16 bits of M0, so a staged block must lie in the first 64 KB of the
    // workgroup's LDS (with the areas behind the range buffers, the waves above 64 KB staged their blocks into other waves' buffers)
    uint32_t* s_x = s_dyn + (size_t)wv * XW;
    uint32_t* s_buf = s_dyn + (size_t)NW * XW + (size_t)wv * BUFW;
    int* ctl = reinterpret_cast<int*>(s_dyn + (size_t)NW * (BUFW + XW));
    static_assert((size_t)NW * XW * 4 <= 65536, "staged blocks within reach of M0");
    ss_sort_image<NW, NEM, NI>(*gp, blk * NI, n_images, keysInAll, keysOutAll, keyCount, maxN, status, nOverride, kthrOverride, depthOverride, s_buf, s_x, ctl,
                               NI == 1 && topAll ? topAll + (size_t)blk * SS_TOP_WORDS : nullptr, grp, NI == 1 ? Gs : 1);
}

// the batch in image groups: NI waves, NI images, the one-wave kernel's block size and LDS per wave -- and its six waves per SIMD
template <int NI>
__global__ __launch_bounds__(64 * NI) __attribute__((amdgpu_waves_per_eu(6, 6)))
void k_lsd_seedsort_grp(const LineGeom* __restrict__ gp, uint32_t* keysInAll, uint32_t* keysOutAll, int* __restrict__ keyCount,
                        const int* __restrict__ maxN, int* __restrict__ status, int nOverride, int kthrOverride, int depthOverride, int n_images)
{
    extern __shared__ __align__(8) uint32_t s_dyn[];
    constexpr int BUFW = SS_CAP, XW = 4 * 64 * SS_NE_LDS;
    const int wv = threadIdx.x >> 6;
    uint32_t* s_x = s_dyn + (size_t)wv * XW;                              // (staging areas first: within reach of M0)
    uint32_t* s_buf = s_dyn + (size_t)NI * XW + (size_t)wv * BUFW;
    int* ctl = reinterpret_cast<int*>(s_dyn + (size_t)NI * (BUFW + XW));
    ss_sort_image<NI, SS_NE_MEM, NI>(*gp, blockIdx.x * NI, n_images, keysInAll, keysOutAll, keyCount, maxN, status, nOverride, kthrOverride, depthOverride, s_buf, s_x, ctl);
}

template <int NI>
static int launch_seedsort_grp(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, int kthrOverride, int depthOverride)
{
    const size_t lds = ((size_t)NI * (SS_CAP + 4 * 64 * SS_NE_LDS) + 4 + 5 * 64 + 24) * 4;
    hipLaunchKernelGGL((k_lsd_seedsort_grp<NI>), dim3((n_images + NI - 1) / NI), dim3(64 * NI), lds, s, b.geom, b.keysA, b.keysB, b.keyCount, b.maxN, b.status,
                       nOverride, kthrOverride, depthOverride, n_images);
    return OLF_OK;
}

// ---- the top of the recursion on the whole GPU (few images) -------------------------------------------------------------------------------------
// With a handful of images the first partitions are what a stereo pair waits for: 668 k, 475 k, 270 k ... elements streamed by ONE wave while the
// rest of the chip idles (a workgroup's waves together are no faster: a single CU keeps ~16 KB in flight).  The ranges of at least SS_TOP_MIN elements
// are therefore partitioned level by level by grid-wide kernels -- one launch per phase, the launch boundary being the barrier -- from the same rank
// formulation as ss_partition: with L(j) the position of the j-th element >= pivot from the left and R(j) that of the j-th element <= pivot from the
// right, the pairs j < s swap, s = #{j : L(j) < R(j)}; f(x) = #{L-stoppers left of x} rises, g(x) = #{R-stoppers at or right of x} falls, and
// s = max_x min(f(x), g(x)) sits where they cross.  Per level: k_top_pivot (median of three, tile tables), k_top_count (stoppers per 64-key tile),
// k_top_scan (f, g at the tile boundaries, crossing tile, s, the cut), k_top_gather (values of the swapping stoppers by rank -- into the seed list's
// memory, which nothing has been emitted to yet), k_top_apply (stored at their partners' positions), k_top_next (children: next level's jobs, or
// entries of the stack the per-image kernel starts from).  Job lists / counters / final entries: b.topBuf; f and g: the image's growth log.
struct SsTop {
    int* cnt; int* jobs; int* next; int* fin;
};
__device__ __forceinline__ SsTop ss_top(int* topAll, int img, int level)
{
    int* b = topAll + (size_t)img * SS_TOP_WORDS;
    SsTop t; t.cnt = b; t.jobs = b + 8 + ((level & 1) ? SS_TOP_JOBS * SS_JW : 0); t.next = b + 8 + ((level & 1) ? 0 : SS_TOP_JOBS * SS_JW); t.fin = b + 8 + 2 * SS_TOP_JOBS * SS_JW;
    return t;
}
__device__ __forceinline__ void ss_top_child(const SsTop& t, int* s_next, int* s_fin, int first, int last, int depth, uint32_t lb, uint32_t ub, uint32_t Kthr, bool lastLevel,
                                             int* status)
{
    if (lb > Kthr) return;                                    // only undefined pixels: never seeds, never leave the range
    // (cnt[6]: the smallest range the grid-wide levels still split -- SS_TOP_MIN, more for a large working image, so that the ranges they leave fit the list)
    if (!lastLevel && last - first >= t.cnt[6] && lb != ub && depth > 0) {
        const int j = atomicAdd(s_next, 1);
        if (j < SS_TOP_JOBS) {
            int* q = t.next + j * SS_JW;
            q[0] = first; q[1] = last; q[2] = depth - 1; q[3] = (int)lb; q[4] = (int)ub;
            return;
        }
        // (more ranges than a level has room for: this one is left to the per-image kernel as it is)
    }
    {
        const int e = atomicAdd(s_fin, 1);
        if (e < SS_TOP_FINAL) { int* q = t.fin + e * 5; q[0] = first; q[1] = last; q[2] = depth; q[3] = (int)lb; q[4] = (int)ub; }
        else atomicOr(status, 64);      // (its own flag: 32 is the frame record buffer)
    }
}

__global__ __launch_bounds__(64) void k_top_init(const LineGeom* __restrict__ gp, int* __restrict__ topAll, const int* __restrict__ maxN, int nOverride, int kthrOverride,
                                                 int depthOverride, int* __restrict__ status, int topMin)
{
    const LineGeom& g = *gp;
    const int img = blockIdx.x;
    if (threadIdx.x) return;
    const SsTop t = ss_top(topAll, img, 0);
    const int n = nOverride >= 0 ? nOverride : (g.Ws - 1) * (g.Hs - 1);
    uint32_t Kthr = 0;
    bool empty = n <= 0;
    if (empty) {}
    else if (kthrOverride >= 0) Kthr = (uint32_t)kthrOverride;
    else {
        const int mN = maxN[img * 32];
        if (mN <= 0) empty = true;
        else {      // (as in ss_sort_image)
            const double max_grad = sqrt((double)mN / 4.0);
            const double bin_coef = (double)(g.nBins - 1) / max_grad;
            const double normT = sqrt((double)g.nThr / 4.0);
            Kthr = (uint32_t)(g.nBins - 1 - (int)(normT * bin_coef));
        }
    }
    t.cnt[0] = 0; t.cnt[1] = 0; t.cnt[2] = 0; t.cnt[3] = 0; t.cnt[4] = (int)Kthr; t.cnt[5] = empty ? 0 : n; t.cnt[6] = topMin;
    if (empty) return;
    const int depth0 = depthOverride >= 0 ? depthOverride : 2 * (31 - __builtin_clz((unsigned)n));
    __shared__ int s_nj, s_nf;
    s_nj = 0; s_nf = 0;
    // (the root through the same rule as every child; the list level 0 reads is the `next` list of a view of "level -1")
    const SsTop tm = ss_top(topAll, img, 1);
    ss_top_child(tm, &s_nj, &s_nf, 0, n, depth0, 0u, (uint32_t)(g.nBins - 1), Kthr, false, status);
    t.cnt[0] = s_nj; t.cnt[2] = s_nf;
}

// one thread per job: __move_median_to_first(first, first + 1, mid, last - 1); thread 0 then numbers the jobs' tiles
// median of three of every job of `level`, moved to the job's first position (std::__move_median_to_first); the level's flat tile numbering
__device__ __forceinline__ void ss_top_pivot_body(int* __restrict__ topAll, uint32_t* keysAll, size_t Ps, int level, int img, int j)
{
    const SsTop t = ss_top(topAll, img, level);
    const int nj = t.cnt[0];
    uint32_t* A = keysAll + (size_t)img * Ps;
    if (j < nj) {
        int* q = t.jobs + j * SS_JW;
        const int first = q[0], last = q[1], mid = first + (last - first) / 2;
        const uint32_t e0 = A[first], ea = A[first + 1], eb = A[mid], ec = A[last - 1];
        const uint32_t Ka = ssK(ea), Kb = ssK(eb), Kc = ssK(ec);
        int sel;
        if (Ka < Kb) sel = Kb < Kc ? 1 : (Ka < Kc ? 2 : 0);
        else sel = Ka < Kc ? 0 : (Kb < Kc ? 2 : 1);
        const uint32_t es = sel == 0 ? ea : sel == 1 ? eb : ec;
        const int sidx = sel == 0 ? first + 1 : sel == 1 ? mid : last - 1;
        A[first] = es; A[sidx] = e0;
        q[5] = (int)ssK(es);
        q[7] = (last - first - 1 + 63) >> 6;
    }
    __syncthreads();
    if (j == 0) {
        int acc = 0;
        for (int k = 0; k < nj; ++k) { int* q = t.jobs + k * SS_JW; q[6] = acc; acc += q[7]; }
        t.cnt[3] = acc;
    }
}
__global__ __launch_bounds__(SS_TOP_JOBS) void k_top_pivot(int* __restrict__ topAll, uint32_t* keysAll, size_t Ps, int level)
{
    ss_top_pivot_body(topAll, keysAll, Ps, level, blockIdx.x, threadIdx.x);
}

// the tile `gt` of the level's flat tile numbering -> its job; PL / GR of job j start at (first tile + j) and leave one spare word per job (T + 1 entries)
__device__ __forceinline__ int ss_top_job_of(const SsTop& t, int nj, int gt)
{
    int j = 0;
    for (int k = 1; k < nj; ++k) if (t.jobs[k * SS_JW + 6] <= gt) j = k;      // (first tiles ascend with the job index)
    return j;
}

template <int PASS>      // 0: count, 1: gather the swapping stoppers' values by rank, 2: store them at their partners' positions
__global__ __launch_bounds__(256) void k_top_tiles(int* __restrict__ topAll, uint32_t* keysAll, uint32_t* outAll, uint32_t* __restrict__ scrAll, size_t Ps, size_t scrStride,
                                                   int half, int level)
{
    const int img = blockIdx.y, lane = threadIdx.x & 63;
    const SsTop t = ss_top(topAll, img, level);
    const int nj = t.cnt[0], gt = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (gt >= t.cnt[3]) return;
    const int j = ss_top_job_of(t, nj, gt);
    const int* q = t.jobs + j * SS_JW;
    const int lo = q[0] + 1, hi = q[1], tt = gt - q[6], s = q[8];
    const uint32_t Kp = (uint32_t)q[5];
    uint32_t* A = keysAll + (size_t)img * Ps;
    uint32_t* PL = scrAll + (size_t)img * scrStride + q[6] + j;
    uint32_t* GR = PL + half;
    const int pos = lo + 64 * tt + lane;
    const bool v = pos < hi;
    if (PASS == 0) {
        const uint32_t e = v ? A[pos] : 0xffffffffu;
        const unsigned long long mL = wave_vote(v && ssK(e) >= Kp), mR = wave_vote(v && ssK(e) <= Kp);
        if (lane == 0) { PL[tt + 1] = (uint32_t)__popcll(mL); GR[tt] = (uint32_t)__popcll(mR); }
        return;
    }
    const int nl0 = (int)PL[tt], nr0 = (int)GR[tt + 1];
    if (nl0 >= s && nr0 >= s) return;
    const uint32_t e = v ? A[pos] : 0xffffffffu;
    const bool isL = v && ssK(e) >= Kp, isR = v && ssK(e) <= Kp;
    const unsigned long long mL = wave_vote(isL), mR = wave_vote(isR);
    const int rl = nl0 + wave_rank_below(mL), rr = nr0 + __popcll((mR >> lane) >> 1);
    // values by rank in the seed list's memory under the range: VL upwards from its start, VR downwards from its end (2 s <= length)
    uint32_t* V = outAll + (size_t)img * Ps;
    if (PASS == 1) {
        if (isL && rl < s) V[lo + rl] = e;
        if (isR && rr < s) V[hi - 1 - rr] = e;
    } else {
        if (isL && rl < s) A[pos] = V[hi - 1 - rl];
        else if (isR && rr < s) A[pos] = V[lo + rr];
    }
}

// one workgroup per job: the counts become f (PL[t] = L-stoppers in tiles < t) and g (GR[t] = R-stoppers in tiles >= t); then s and the cut
constexpr int SS_SCAN_NT = 1024;      // (a job of the first level has ten thousand tiles: 1024 threads walk eleven each, 256 walked forty-one -- 35 -> 15 us)
__global__ __launch_bounds__(SS_SCAN_NT) void k_top_scan(int* __restrict__ topAll, const uint32_t* __restrict__ keysAll, uint32_t* __restrict__ scrAll, size_t Ps, size_t scrStride,
                                                  int half, int level)
{
    constexpr int NT = SS_SCAN_NT;
    __shared__ int s_red[2 * NT];
    const int img = blockIdx.y, j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SsTop t = ss_top(topAll, img, level);
    if (j >= t.cnt[0]) return;
    int* q = t.jobs + j * SS_JW;
    const int lo = q[0] + 1, hi = q[1], T = q[7];
    const uint32_t Kp = (uint32_t)q[5];
    const uint32_t* A = keysAll + (size_t)img * Ps;
    uint32_t* PL = scrAll + (size_t)img * scrStride + q[6] + j;
    uint32_t* GR = PL + half;
    const int C = (T + NT - 1) / NT, c0 = min(tid * C, T), c1 = min(c0 + C, T);
    {
        int sl = 0, sr = 0;
        for (int x = c0; x < c1; ++x) { sl += (int)PL[x + 1]; sr += (int)GR[x]; }
        s_red[tid] = sl; s_red[NT + tid] = sr;
    }
    __syncthreads();
    if (wv == 0) {
        int accL = 0, accR = 0;
        for (int b = 0; b < NT / 64; ++b) {                   // prefix over the threads' chunk sums (left to right) and suffix (right to left)
            const int i = b * 64 + lane, k = (NT / 64 - 1 - b) * 64 + (63 - lane);
            const int vl = s_red[i], vr = s_red[NT + k];
            int pl = vl, pr = vr;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int xl = __shfl_up(pl, o), xr = __shfl_up(pr, o); if (lane >= o) { pl += xl; pr += xr; } }
            const int totL = __shfl(pl, 63), totR = __shfl(pr, 63);
            s_red[i] = accL + pl - vl;
            s_red[NT + k] = accR + pr - vr;
            accL += totL; accR += totR;
        }
    }
    __syncthreads();
    {
        int run = s_red[tid];
        if (tid == 0) PL[0] = 0;
        for (int x = c0; x < c1; ++x) { run += (int)PL[x + 1]; PL[x + 1] = (uint32_t)run; }
        run = s_red[NT + tid];
        for (int x = c1 - 1; x >= c0; --x) { run += (int)GR[x]; GR[x] = (uint32_t)run; }
        if (tid == 0) GR[T] = 0;
    }
    __threadfence_block();
    __syncthreads();
    if (wv != 0) return;
    // the crossing: the first boundary tb with f(tb) >= g(tb); s = max min(f, g) over the positions of tile tb - 1 and its two boundaries
    int a = 0, b = T;                                         // f(0) = 0 <= g(0); f(T) = NL >= 0 = g(T)
    while (a < b) { const int m = (a + b) >> 1; if (PL[m] >= GR[m]) b = m; else a = m + 1; }
    const int tb = a;
    int s = (int)min(PL[tb], GR[tb]);
    if (tb > 0) {
        const int u = tb - 1, pos = lo + 64 * u + lane;
        const bool v = pos < hi;
        const uint32_t e = v ? A[pos] : 0xffffffffu;
        const unsigned long long mL = wave_vote(v && ssK(e) >= Kp), mR = wave_vote(v && ssK(e) <= Kp);
        const int f = (int)PL[u] + wave_rank_below(mL);                                     // L-stoppers left of this position
        const int gq = (int)GR[u + 1] + __popcll(mR >> lane);                               // R-stoppers at or right of it
        int best = min(f, gq);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
        s = max(s, best);
    }
    // cut = min(L(s), R(s - 1)) over the ones that exist, else hi
    const int NL = (int)PL[T], NR = (int)GR[0];
    int cut = hi;
    if (s < NL) {                                             // L(s): tile x with PL[x] <= s < PL[x + 1]
        int x = 0, y = T - 1;
        while (x < y) { const int m = (x + y + 1) >> 1; if ((int)PL[m] <= s) x = m; else y = m - 1; }
        const int pos = lo + 64 * x + lane;
        const bool v = pos < hi;
        const uint32_t e = v ? A[pos] : 0xffffffffu;
        const unsigned long long mL = wave_vote(v && ssK(e) >= Kp);
        const unsigned long long hit = wave_vote(v && ssK(e) >= Kp && (int)PL[x] + wave_rank_below(mL) == s);
        if (hit) cut = min(cut, lo + 64 * x + (int)__builtin_ctzll(hit));
    }
    if (s >= 1 && s - 1 < NR) {                               // R(s - 1): tile x with GR[x + 1] <= s - 1 < GR[x]
        int x = 0, y = T - 1;
        while (x < y) { const int m = (x + y) >> 1; if ((int)GR[m + 1] <= s - 1) y = m; else x = m + 1; }
        const int pos = lo + 64 * x + lane;
        const bool v = pos < hi;
        const uint32_t e = v ? A[pos] : 0xffffffffu;
        const unsigned long long mR = wave_vote(v && ssK(e) <= Kp);
        const unsigned long long hit = wave_vote(v && ssK(e) <= Kp && (int)GR[x + 1] + __popcll((mR >> lane) >> 1) == s - 1);
        if (hit) cut = min(cut, lo + 64 * x + (int)__builtin_ctzll(hit));
    }
    if (lane == 0) { q[8] = s; q[9] = cut; }
}

// children of every job: [cut, last) with K >= Kp, [first, cut) with K <= Kp (the pivot sits at first)
// ... and, unless this was the last level, the pivots of the next level's jobs in the same launch (one launch boundary less per level)
__global__ __launch_bounds__(SS_TOP_JOBS) void k_top_next(int* __restrict__ topAll, int level, int* __restrict__ status, uint32_t* keysAll, size_t Ps)
{
    __shared__ int s_next, s_fin;
    const int img = blockIdx.x, j = threadIdx.x;
    const SsTop t = ss_top(topAll, img, level);
    const int nj = t.cnt[0];
    if (j == 0) { s_next = 0; s_fin = t.cnt[2]; }
    __syncthreads();
    if (j < nj) {
        const int* q = t.jobs + j * SS_JW;
        const int first = q[0], last = q[1], depth = q[2], cut = q[9];
        const uint32_t lb = (uint32_t)q[3], ub = (uint32_t)q[4], Kp = (uint32_t)q[5], Kthr = (uint32_t)t.cnt[4];
        const bool lastLevel = level == SS_TOP_LEVELS - 1;
        ss_top_child(t, &s_next, &s_fin, cut, last, depth, max(lb, Kp), ub, Kthr, lastLevel, status);
        ss_top_child(t, &s_next, &s_fin, first, cut, depth, lb, min(ub, Kp), Kthr, lastLevel, status);
    }
    __syncthreads();
    if (j == 0) { t.cnt[0] = min(s_next, SS_TOP_JOBS); t.cnt[2] = min(s_fin, SS_TOP_FINAL); t.cnt[3] = 0; }
    if (level == SS_TOP_LEVELS - 1) return;
    __syncthreads();
    ss_top_pivot_body(topAll, keysAll, Ps, level + 1, img, j);
}

int lsd_seedsort_top_words() { return SS_TOP_WORDS; }

static int launch_seedsort_top(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, int kthrOverride, int depthOverride)
{
    const int n = nOverride >= 0 ? nOverride : (g.Ws - 1) * (g.Hs - 1);
    const int maxTiles = n / 64 + SS_TOP_JOBS + 2, half = maxTiles + SS_TOP_JOBS + 2;
    const size_t stride = (size_t)g.regionStride;
    if ((size_t)2 * half > stride) return OLF_ERR_CAPACITY;
    // ranges of at least topMin elements are split by the grid-wide levels: SS_TOP_MIN, or more when the image is so large that every pixel defined (noise) would
    // leave more than about 160 ranges (the list holds SS_TOP_FINAL = 512, a level SS_TOP_JOBS = 128 jobs)
    int topMin = SS_TOP_MIN;
    while (n / topMin > 160) topMin <<= 1;
    hipLaunchKernelGGL(k_top_init, dim3(n_images), dim3(64), 0, s, b.geom, b.topBuf, b.maxN, nOverride, kthrOverride, depthOverride, b.status, topMin);
    if (n < topMin) return OLF_OK;       // (the root is a final entry)
    const dim3 tg((maxTiles + 3) / 4, n_images);
    for (int level = 0; level < SS_TOP_LEVELS; ++level) {
        if (level == 0) hipLaunchKernelGGL(k_top_pivot, dim3(n_images), dim3(SS_TOP_JOBS), 0, s, b.topBuf, b.keysA, (size_t)g.Ps, level);
        hipLaunchKernelGGL(k_top_tiles<0>, tg, dim3(256), 0, s, b.topBuf, b.keysA, b.keysB, b.region, (size_t)g.Ps, stride, half, level);
        hipLaunchKernelGGL(k_top_scan, dim3(SS_TOP_JOBS, n_images), dim3(SS_SCAN_NT), 0, s, b.topBuf, b.keysA, b.region, (size_t)g.Ps, stride, half, level);
        hipLaunchKernelGGL(k_top_tiles<1>, tg, dim3(256), 0, s, b.topBuf, b.keysA, b.keysB, b.region, (size_t)g.Ps, stride, half, level);
        hipLaunchKernelGGL(k_top_tiles<2>, tg, dim3(256), 0, s, b.topBuf, b.keysA, b.keysB, b.region, (size_t)g.Ps, stride, half, level);
        hipLaunchKernelGGL(k_top_next, dim3(n_images), dim3(SS_TOP_JOBS), 0, s, b.topBuf, level, b.status, b.keysA, (size_t)g.Ps);
    }
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

template <int NW, int NEM, int NI>
static int launch_seedsort_mw(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, int kthrOverride, int depthOverride)
{
    // per wave: range buffer + staging area; then lock / counters, the shared stack's five arrays (256 entries, 64 for image groups) and the groups' slots
    const size_t lds = ((size_t)NW * (4 * 64 * NEM + 2 * 64 * NEM) + 4 + 5 * (NI > 1 ? 64 : 256) + 24) * 4;
    if (lds > 64 * 1024) {     // the attribute belongs to the device the launch goes to: set once per device (the call is a host round trip in front of every one-pair call otherwise)
        static bool done[64] = {};
        int dev = 0;
        OLF_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !done[dev]) {
            OLF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsd_seedsort_mw<NW, NEM, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) done[dev] = true;
        }
    }
    // OLF_SS_TOP=0: the whole recursion inside the per-image workgroup (A/B measurements)
    static const bool top = [] { const char* e = getenv("OLF_SS_TOP"); return !e || atoi(e) != 0; }();
    // (few images only: the grids cover every possible tile of every image at every level -- at 128 images the two forms are level, at 1024 the
    // grid-wide one loses 104 against 73 ms, on a 1080p batch 330 against 102)
    const bool useTop = top && NI == 1 && b.topBuf && n_images <= 64;
    if (useTop) { const int rc = launch_seedsort_top(g, b, n_images, s, nOverride, kthrOverride, depthOverride); if (rc != OLF_OK) return rc; }
    // behind the top levels a few images leave most of the chip idle: Gs workgroups (CUs) per image, each starting from every Gs-th of the ranges the top levels
    // left (OLF_SS_GROUPS forces 1 .. 8)
    static const int envG = [] { const char* e = getenv("OLF_SS_GROUPS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 8 ? v : 0; }();
    const int Gs = !useTop ? 1 : envG ? envG : n_images <= 16 ? 8 : 4;      // (useTop: at most 64 images; 8 pairs 9.85 against 9.91 ms with 4, 32 pairs 14.0 with 4 against 14.6 with 8)
    // (the groups' seed counts meet in an atomicMax: the counts start at zero -- launch_lsd_front has cleared them; the debug entry, which comes without a front, has not)
    if (Gs > 1 && nOverride >= 0) OLF_HIP_CHECK(hipMemsetAsync(b.keyCount, 0, (size_t)n_images * 32 * sizeof(int), s));
    hipLaunchKernelGGL((k_lsd_seedsort_mw<NW, NEM, NI>), dim3(((n_images + NI - 1) / NI) * Gs), dim3(64 * NW), lds, s, b.geom, b.keysA, b.keysB, b.keyCount, b.maxN, b.status,
                       nOverride, kthrOverride, depthOverride, n_images, useTop ? b.topBuf : (int*)nullptr, Gs);
    return OLF_OK;
}

int launch_lsd_seedsort(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, int kthrOverride, int depthOverride)
{
    // OLF_SS_MW: 0 forces the one-wave kernel, 1 / 2 the 4- / 8-wave variant, 3 / 4 groups of 4 / 8 images (A/B measurements); default: by batch size
    static const int forced = [] { const char* e = getenv("OLF_SS_MW"); return e ? atoi(e) : -1; }();
    // up to 256 images: 8 waves per image (101 KB of LDS, one workgroup per CU); up to 640: 4 waves (53 KB, three per CU); up to 1536: 2 waves (30 KB, five per
    // CU: the 1280 images of a 1080p batch go 144 -> 103 ms; KITTI size, ms: 512 images 9.3 / 12.2 with 4 / 2 waves, 768: 17.3 / 13.0, 1024: 18.3 / 13.7 and
    // 23.2 with one, 1536: 27.1 / 24.2 / 24.7); beyond: one wave per image --
    // alone (mode 0).  Modes 3 / 4 (groups of 4 / 8 images whose waves take over each other's streamed ranges) are opt-in: on 6144 copies of 32 images
    // the kernel goes 43.5 -> 37.9 ms (groups of 8; 40.2 with 4; equal at 1536 images), on the bench's 512 distinct pairs the front does not move
    // (71.3 against 71.7 ms) and the step is 278.9 against 276.8 ms -- the launch is bound by issue slots, not by its slowest image.
    // One stereo pair through olf_stereo_frames, host to host: 25.6 ms with the one-wave kernel, 16.5 ms with 4 waves, 15.1 ms with 8
    int mode = b.forceSortMode >= 0 ? b.forceSortMode : forced >= 0 ? forced : (n_images <= 256 ? 2 : n_images <= 640 ? 1 : n_images <= 1536 ? 5 : 0);
    {   // the multi-wave kernels ask for 101 / 53 / 30 KB of dynamic LDS (8 / 4 / 2 waves): on a device whose workgroups cannot have that much (the Makefile
        // accepts other ARCH values than gfx950) take the largest variant that fits instead of failing the launch -- the result does not depend on it
        int dev = 0, maxLds = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&maxLds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) maxLds = 64 * 1024;
        auto need = [](int m) { return m == 2 ? 104 * 1024 : m == 1 ? 56 * 1024 : m == 5 ? 32 * 1024 : m == 4 ? 104 * 1024 : m == 3 ? 56 * 1024 : 0; };
        while (need(mode) > maxLds) mode = mode == 2 ? 1 : mode == 1 ? 5 : mode == 4 ? 3 : 0;
    }
    int rc = OLF_OK;
    if (mode == 1) rc = launch_seedsort_mw<4, 8, 1>(g, b, n_images, s, nOverride, kthrOverride, depthOverride);
    else if (mode == 2) rc = launch_seedsort_mw<8, 8, 1>(g, b, n_images, s, nOverride, kthrOverride, depthOverride);
    else if (mode == 5) rc = launch_seedsort_mw<2, 8, 1>(g, b, n_images, s, nOverride, kthrOverride, depthOverride);
    else if (mode == 3) rc = launch_seedsort_grp<4>(g, b, n_images, s, nOverride, kthrOverride, depthOverride);
    else if (mode == 4) rc = launch_seedsort_grp<8>(g, b, n_images, s, nOverride, kthrOverride, depthOverride);
    else
        hipLaunchKernelGGL(k_lsd_seedsort, dim3(n_images), dim3(64), 0, s, b.geom, b.keysA, b.keysB, b.keyCount, b.maxN, b.status, nOverride, kthrOverride, depthOverride);
    if (rc != OLF_OK) return rc;
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
