// bow_match.hip -- ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:161-290) for a whole
// batch of consecutive frames on the device, Frame::ComputeBoW (src/Frame.cc:585-597) included: the matcher BASELINE.json's configuration 3 names
// ("SearchByBoW match vs prev KF"), with no host step between the descriptors of a frame and its matches.
//
//   k_bow_descend_nodes  TemplatedVocabulary::transform(feature, word, weight, nid, levelsup) per descriptor (Thirdparty/DBoW2/DBoW2/
//                        TemplatedVocabulary.h:1217-1261): the node at `levelsup` levels above the leaves; -1 when the word's weight is 0
//                        (transform() leaves such a feature out of the FeatureVector, :1165-1172)
//   k_bow_sort_nodes     per frame: the FeatureVector as one sorted list of (node << 16 | feature index) -- node ids ascending like the std::map,
//                        feature indices ascending inside a node like the vectors addFeature() appends to (bitonic sort in LDS)
//   k_search_by_bow      per pair (key frame = frame j, frame = frame j + 1): the nodes both lists share are independent of each other (a feature
//                        belongs to one node), so the waves of a workgroup take them one at a time; INSIDE a node the reference's greedy state --
//                        a feature of F that already holds a match is skipped (:214-215) -- makes the key frame's features sequential: one
//                        wave walks them in index order, its lanes hold the node's features of F, best / second-best distance by wave reductions
//                        with the reference's scan-order tie rule; then the rotation histogram (ComputeThreeMaxima, :1749-1790) per pair.
#include "olf_internal.hpp"

namespace olf {

constexpr int BM_TH_LOW = 50, BM_HISTO = 30;      // src/ORBmatcher.cc:39-41

__global__ __launch_bounds__(256) void k_bow_descend_nodes(const uint4* __restrict__ slotDesc, const int* __restrict__ childOff, const int* __restrict__ slotNode,
                                                           const double* __restrict__ nodeWeight, const uint4* __restrict__ desc, const int* __restrict__ counts,
                                                           int cap, int img_stride, int nid_level, int* __restrict__ nodeOut)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    int nid = -1;
    if (i < counts[(size_t)f * img_stride]) {
        const uint4* d = desc + 2 * ((size_t)f * img_stride * cap + i);
        const uint4 f0 = d[0], f1 = d[1];
        int node = 0, level = 0;
        int b = childOff[0], e = childOff[1];
        while (e > b) {
            ++level;
            int best = b, bd = 257;
            for (int s = b; s < e; ++s) {
                const uint4 c0 = slotDesc[2 * (size_t)s], c1 = slotDesc[2 * (size_t)s + 1];
                const int dd = __popc(f0.x ^ c0.x) + __popc(f0.y ^ c0.y) + __popc(f0.z ^ c0.z) + __popc(f0.w ^ c0.w) + __popc(f1.x ^ c1.x) +
                               __popc(f1.y ^ c1.y) + __popc(f1.z ^ c1.z) + __popc(f1.w ^ c1.w);
                if (dd < bd) { bd = dd; best = s; }
            }
            node = slotNode[best];
            if (level == nid_level) nid = node;
            b = childOff[node]; e = childOff[node + 1];
        }
        if (!(nodeWeight[node] > 0)) nid = -1;      // if (w > 0) fv.addFeature(nid, i_feature), TemplatedVocabulary.h:1165-1172
        else if (nid < 0) nid = 0;                  // a tree shallower than nid_level never sets nid: DBoW2 leaves it 0 (the root)
    }
    nodeOut[(size_t)f * cap + i] = nid;
}

// sorted[f][0 .. m[f]) = (node << 16 | index) ascending, P = power of two >= cap entries of LDS
__global__ __launch_bounds__(256) void k_bow_sort_nodes(const int* __restrict__ nodeIn, int cap, int P, unsigned long long* __restrict__ sorted, int* __restrict__ mOut)
{
    extern __shared__ unsigned long long s_key[];
    __shared__ int s_m;
    const int f = blockIdx.x;
    if (threadIdx.x == 0) s_m = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < P; i += 256) {
        const int nd = i < cap ? nodeIn[(size_t)f * cap + i] : -1;
        s_key[i] = nd >= 0 ? ((unsigned long long)nd << 16) | (unsigned)i : ~0ull;
        mine += nd >= 0;
    }
    if (mine) atomicAdd(&s_m, mine);
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += 256) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool up = (lo & k) == 0;
                const unsigned long long a = s_key[lo], b = s_key[hi];
                if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < cap; i += 256) sorted[(size_t)f * cap + i] = s_key[i];
    if (threadIdx.x == 0) mOut[f] = s_m;
}

#ifndef OLF_BM_WAVES
#define OLF_BM_WAVES 8
#endif
constexpr int BM_WAVES = OLF_BM_WAVES;

__device__ __forceinline__ int bm_lower_bound(const unsigned long long* a, int n, unsigned long long key)      // first position with a[p] >= key
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// (BM_WAVES waves per frame pair: the pair's ~100 shared nodes are claimed one at a time by whichever wave is free; the walk inside a node is serial, so the
// kernel's time is the longest chain of nodes one wave ends up with: 4 waves 3.60 ms per 3071 pairs, 8 waves 3.07, 16 waves 4.72 -- profiles/r4at_bow_waves_ab.txt)
__global__ __launch_bounds__(64 * BM_WAVES) void k_search_by_bow(const unsigned long long* __restrict__ sortedAll, const int* __restrict__ mAll, const olf_keypoint* __restrict__ kps,
                                                       const uint4* __restrict__ desc, const int* __restrict__ counts, int cap, int img_stride,
                                                       const uint8_t* __restrict__ mpValid, const uint8_t* __restrict__ mpBad, float nnratio, int checkOri,
                                                       int* __restrict__ matches, int* __restrict__ nmatches)
{
    extern __shared__ int s_mem[];                   // matched[cap] (key-frame feature or -1), then one rotation bin byte per feature
    __shared__ int s_hist[BM_HISTO], s_seg, s_n, s_keep[3];
    int* matched = s_mem;
    uint8_t* binOf = reinterpret_cast<uint8_t*>(s_mem + cap);
    const int p = blockIdx.x, lane = threadIdx.x & 63;
    const int fK = p, fF = p + 1;
    const unsigned long long* SK = sortedAll + (size_t)fK * cap;
    const unsigned long long* SF = sortedAll + (size_t)fF * cap;
    const int mK = mAll[fK], mF = mAll[fF], nF = counts[(size_t)fF * img_stride];
    const olf_keypoint* kK = kps + (size_t)fK * img_stride * cap;
    const olf_keypoint* kF = kps + (size_t)fF * img_stride * cap;
    const uint4* dK = desc + 2 * (size_t)fK * img_stride * cap;
    const uint4* dF = desc + 2 * (size_t)fF * img_stride * cap;
    for (int i = threadIdx.x; i < cap; i += 64 * BM_WAVES) { matched[i] = -1; binOf[i] = 0; }
    if (threadIdx.x < BM_HISTO) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_seg = 0; s_n = 0; }
    __syncthreads();
    // the key frame's list is cut into node segments on the fly: a wave claims the next unclaimed list position, finds the end of the node it
    // starts (s_seg always sits on a segment head) and moves s_seg there
    for (;;) {
        int kb = 0, ke = 0;
        if (lane == 0) {
            // claim [kb, ke): compare-and-swap so that exactly one wave advances the head from kb to ke
            for (;;) {
                kb = atomicAdd(&s_seg, 0);
                if (kb >= mK) { ke = kb; break; }
                const unsigned long long node = SK[kb] >> 16;
                ke = bm_lower_bound(SK, mK, (node + 1) << 16);
                if (atomicCAS(&s_seg, kb, ke) == kb) break;
            }
        }
        kb = __builtin_amdgcn_readfirstlane(kb); ke = __builtin_amdgcn_readfirstlane(ke);
        if (kb >= mK) break;
        const unsigned long long node = SK[kb] >> 16;
        const int fb = bm_lower_bound(SF, mF, node << 16), fe = bm_lower_bound(SF, mF, (node + 1) << 16);
        if (fe <= fb) continue;
        for (int q = kb; q < ke; ++q) {
            const int iKF = (int)(SK[q] & 0xffffu);
            if (mpValid && !mpValid[(size_t)fK * cap + iKF]) continue;
            if (mpBad && mpBad[(size_t)fK * cap + iKF]) continue;
            const uint4 a0 = dK[2 * (size_t)iKF], a1 = dK[2 * (size_t)iKF + 1];
            int b1 = 256, bi = -1, b2 = 256;
            for (int c0 = fb; c0 < fe; c0 += 64) {
                const bool on = c0 + lane < fe;
                const int iF = on ? (int)(SF[c0 + lane] & 0xffffu) : 0;
                int d = 0x7fff;
                if (on && matched[iF] < 0) {
                    const uint4 x0 = dF[2 * (size_t)iF], x1 = dF[2 * (size_t)iF + 1];
                    d = __popc(a0.x ^ x0.x) + __popc(a0.y ^ x0.y) + __popc(a0.z ^ x0.z) + __popc(a0.w ^ x0.w) + __popc(a1.x ^ x1.x) + __popc(a1.y ^ x1.y) +
                        __popc(a1.z ^ x1.z) + __popc(a1.w ^ x1.w);
                }
                // smallest (distance, lane) of the chunk, then the smallest distance among the other lanes
                // (DPP minima with a scalar result: the two 6-step butterflies through the LDS crossbar were the dependent chain of this serial walk)
                const int key = wave_min_i32((d << 6) | lane);
                const int c1 = key >> 6, cl = key & 63;
                const int d2 = wave_min_i32(lane == cl ? 0x7fff : d);
                if (c1 < 0x7fff) {
                    // the chunk's candidates come after the earlier chunks' in the reference's scan: `<` keeps the earlier one on a tie
                    const int ci = __builtin_amdgcn_readlane(iF, cl);
                    if (c1 < b1) { b2 = min(b1, min(d2, 256)); b1 = c1; bi = ci; }
                    else b2 = min(b2, c1);
                }
            }
            if (b1 <= BM_TH_LOW && static_cast<float>(b1) < nnratio * static_cast<float>(b2)) {
                if (lane == 0) {
                    matched[bi] = iKF;
                    if (checkOri) {
                        float rot = kK[iKF].angle - kF[bi].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * (1.0f / BM_HISTO));
                        if (bin == BM_HISTO) bin = 0;
                        binOf[bi] = (uint8_t)bin;
                        atomicAdd(&s_hist[bin], 1);
                    }
                    atomicAdd(&s_n, 1);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    if (checkOri) {
        if (threadIdx.x == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;      // ComputeThreeMaxima, src/ORBmatcher.cc:1749-1790
            for (int i = 0; i < BM_HISTO; i++) {
                const int s = s_hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) ind3 = -1;
            s_keep[0] = ind1; s_keep[1] = ind2; s_keep[2] = ind3;
        }
        __syncthreads();
        int dropped = 0;
        for (int i = threadIdx.x; i < nF; i += 64 * BM_WAVES)
            if (matched[i] >= 0) { const int b = binOf[i]; if (b != s_keep[0] && b != s_keep[1] && b != s_keep[2]) { matched[i] = -1; ++dropped; } }
        if (dropped) atomicSub(&s_n, dropped);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < cap; i += 64 * BM_WAVES) matches[(size_t)p * cap + i] = i < nF ? matched[i] : -1;
    if (threadIdx.x == 0) nmatches[p] = s_n;
}

int launch_search_by_bow_batch(const uint8_t* slotDesc, const int* childOff, const int* slotNode, const double* nodeWeight, int nid_level, int n_frames,
                               int img_stride, int cap, const olf_keypoint* d_kps, const uint8_t* d_desc, const int* d_counts, const uint8_t* d_mp_valid,
                               const uint8_t* d_mp_bad, float nnratio, int check_ori, int* d_nodes, unsigned long long* d_sorted, int* d_m, int* d_matches,
                               int* d_nmatches, hipStream_t s)
{
    if (n_frames < 2) return OLF_OK;
    int P = 64;
    while (P < cap) P <<= 1;
    if (cap > 4096) { set_error("olf_search_by_bow_batch_dev: more than 4096 features per frame (the per-frame node sort runs in 32 KB of LDS)"); return OLF_ERR_CAPACITY; }
    hipLaunchKernelGGL(k_bow_descend_nodes, dim3((cap + 255) / 256, n_frames), dim3(256), 0, s, reinterpret_cast<const uint4*>(slotDesc), childOff, slotNode,
                       nodeWeight, reinterpret_cast<const uint4*>(d_desc), d_counts, cap, img_stride, nid_level, d_nodes);
    hipLaunchKernelGGL(k_bow_sort_nodes, dim3(n_frames), dim3(256), (size_t)P * 8, s, d_nodes, cap, P, d_sorted, d_m);
    hipLaunchKernelGGL(k_search_by_bow, dim3(n_frames - 1), dim3(64 * BM_WAVES), (size_t)cap * 4 + ((cap + 3) & ~3), s, d_sorted, d_m, d_kps,
                       reinterpret_cast<const uint4*>(d_desc), d_counts, cap, img_stride, d_mp_valid, d_mp_bad, nnratio, check_ori, d_matches, d_nmatches);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
