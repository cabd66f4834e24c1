// bow.hip -- DBoW2 vocabulary-tree descent on gfx950.
//   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1261
// called once per descriptor by Frame::ComputeBoW (src/Frame.cc:585-597) for the ORB and the LBD vocabulary.
// One thread per descriptor, the 256-bit feature in registers; the children of a node are stored contiguously (slot order = the
// reference's children vector order), so a level is one 32*k-byte contiguous read.  The top levels (1 + k + k^2 + ... nodes) stay
// in L2; the leaves' level is the only one that misses.  Strict '<' keeps the first minimum, like the reference's scan.
#include "olf_internal.hpp"

namespace olf {

__global__ __launch_bounds__(256) void k_bow_descend(const uint4* __restrict__ slotDesc, const int* __restrict__ childOff,
                                                     const int* __restrict__ slotNode, const int* __restrict__ nodeWord,
                                                     const double* __restrict__ nodeWeight, const uint4* __restrict__ desc, int n, int nid_level,
                                                     int* __restrict__ word, double* __restrict__ weight, int* __restrict__ nodeOut)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 f0 = desc[2 * (size_t)i], f1 = desc[2 * (size_t)i + 1];
    int node = 0, level = 0, nid = 0;
    int b = childOff[0], e = childOff[1];
    while (e > b) {
        ++level;
        int best = b, bd = 257;
        for (int s = b; s < e; ++s) {
            const uint4 c0 = slotDesc[2 * (size_t)s], c1 = slotDesc[2 * (size_t)s + 1];
            const int d = __popc(f0.x ^ c0.x) + __popc(f0.y ^ c0.y) + __popc(f0.z ^ c0.z) + __popc(f0.w ^ c0.w) + __popc(f1.x ^ c1.x) +
                          __popc(f1.y ^ c1.y) + __popc(f1.z ^ c1.z) + __popc(f1.w ^ c1.w);
            if (d < bd) { bd = d; best = s; }
        }
        node = slotNode[best];
        if (level == nid_level) nid = node;
        b = childOff[node]; e = childOff[node + 1];
    }
    word[i] = nodeWord[node];
    weight[i] = nodeWeight[node];
    nodeOut[i] = nid;
}

int launch_bow_descend(const uint8_t* slotDesc, const int* childOff, const int* slotNode, const int* nodeWord, const double* nodeWeight,
                       const uint8_t* desc, int n, int nid_level, int* word, double* weight, int* nodeOut, hipStream_t s)
{
    if (n <= 0) return OLF_OK;
    hipLaunchKernelGGL(k_bow_descend, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const uint4*>(slotDesc), childOff, slotNode, nodeWord,
                       nodeWeight, reinterpret_cast<const uint4*>(desc), n, nid_level, word, weight, nodeOut);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
