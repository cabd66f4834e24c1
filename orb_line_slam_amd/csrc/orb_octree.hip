// orb_octree.hip -- ORBextractor::DistributeOctTree (reference src/ORBextractor.cc:541-765,
// ExtractorNode::DivideNode :483-539) as one 256-thread workgroup per (image, level).
//
// The reference walks a std::list sequentially; its result (which leaves exist, and in which list
// order) is reproduced here with data-parallel steps.  One "round" of the reference is
//   * a full pass over the list (every multi-point node divided, children pushed to the FRONT of
//     the list in the order n1..n4, parent erased), or
//   * once size + 3*nToExpand > N, a "careful" pass: the expandable nodes sorted by
//     (point count, creation order) and divided largest first until the list holds N nodes.
// Both are the same operation with a different processing order pi over the expandable nodes and
// a cut-off K: new list = children of pi[K-1],...,pi[0] (each as n4,n3,n2,n1, empty ones dropped)
// followed by the untouched nodes in their old order.  Processing order, cut-off and list
// positions come from block-wide prefix sums (and one bitonic sort in careful rounds); key points
// only carry the id of the node they sit in, so no per-node vectors are moved.
// Convention C.1: equal counts are expanded most-recently-created first.
#include "olf_internal.hpp"

namespace olf {

struct NodeRec { short ulx, uly, urx, bry; };

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan of a[0..n) in place (n <= 16*256); returns the total.  256 threads, 4 waves.
__device__ int block_excl_scan(int* a, int n, int* wsum)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = (n + 255) >> 8;
    const int b = tid * chunk, e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; ++i) s += a[i];
    int inc = wave_incl_scan(s, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int run = base + inc - s;
    for (int i = b; i < e; ++i) { int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

__device__ __forceinline__ int quadrant(const NodeRec& n, int x, int y, int& hx, int& hy)
{
    hx = (int)ceilf((float)(n.urx - n.ulx) / 2);
    hy = (int)ceilf((float)(n.bry - n.uly) / 2);
    // n1=0 (left,top) n2=1 (right,top) n3=2 (left,bottom) n4=3 (right,bottom)
    return (x < n.ulx + hx ? 0 : 1) + (y < n.uly + hy ? 0 : 2);
}

// SPILL: the working arrays of a level with more than 2048 nodes (more than 2040 key points on one level: nfeatures in the tens of
// thousands) do not fit the 160 KB of LDS; that instance keeps them in a per-(image, level) slice of global memory instead -- same code, the
// workgroup is on one CU, so __syncthreads() orders its global accesses like its LDS ones.  No reference configuration comes near it.
template <bool SPILL>
__global__ __launch_bounds__(256) void k_octree(const OrbGeom* __restrict__ gp, const uint32_t* __restrict__ cells,
                                                const int* __restrict__ cellCount, uint32_t* __restrict__ cand,
                                                uint16_t* __restrict__ candNode, int* __restrict__ candCount,
                                                uint32_t* __restrict__ lvlKp, int* __restrict__ lvlCount, int* __restrict__ status,
                                                unsigned char* __restrict__ spill, size_t spillBytes)
{
    OLF_SET_GUEST_PRIO();
    extern __shared__ __align__(16) unsigned char lds_smem[];
    unsigned char* smem = SPILL ? spill + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * spillBytes : lds_smem;
    const OrbGeom& g = *gp;
    const int level = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const LevelGeom& L = g.lv[level];
    const int M = g.maxNodes;
    // LDS carve-up
    NodeRec* nodeA = reinterpret_cast<NodeRec*>(smem);            // M
    NodeRec* nodeB = nodeA + M;                                     // M
    int* cntA = reinterpret_cast<int*>(nodeB + M);                  // M
    int* cntB = cntA + M;                                           // M
    int* childCnt = cntB + M;                                       // 4M
    int* scanA = childCnt + 4 * M;                                  // M   (rank among expandable / scratch)
    int* scanB = scanA + M;                                         // M   (nc in processing order -> exclusive scan)
    int* scanC = scanB + M;                                         // M   (survivor flags -> exclusive scan)
    uint32_t* sortKey = reinterpret_cast<uint32_t*>(scanC + M);     // M
    unsigned short* seqA = reinterpret_cast<unsigned short*>(sortKey + M);   // M  creation index of node (cur list)
    unsigned short* seqB = seqA + M;                                // M
    unsigned short* procNode = seqB + M;                            // M  processing order -> node
    unsigned short* rankOf = procNode + M;                          // M  node -> rank (0xffff: not processed)
    unsigned short* seqToNode = rankOf + M;                         // M
    int* cellScan = reinterpret_cast<int*>(seqToNode + M);          // 4096 (cells of a level, scanned in chunks)
    __shared__ int wsum[4];
    __shared__ int sh[8];

    const int nCells = L.nCols * L.nRows;
    const int* cc = cellCount + (size_t)img * g.totalCells + L.cellBase;
    uint32_t* myCand = cand + (size_t)img * g.candTotal + L.candBase;
    uint16_t* myNode = candNode + (size_t)img * g.candTotal + L.candBase;

    // ---- gather the cells' candidates in reference order (cell-row, cell-col, row-major in cell)
    int C = 0;
    for (int base = 0; base < nCells; base += 4096) {
        const int n = min(4096, nCells - base);
        for (int i = tid; i < n; i += 256) cellScan[i] = cc[base + i];
        __syncthreads();
        const int tot = block_excl_scan(cellScan, n, wsum);
        for (int i = tid; i < n; i += 256) {
            const int k = cc[base + i], o = C + cellScan[i];
            const uint32_t* slot = cells + ((size_t)img * g.totalCells + L.cellBase + base + i) * g.cellCap;
            for (int j = 0; j < k; ++j)
                if (o + j < L.candCap) myCand[o + j] = slot[j];
        }
        C += tot;
        __syncthreads();
    }
    if (C > L.candCap) { if (tid == 0) atomicOr(status, 1); C = L.candCap; }
    if (tid == 0) candCount[img * g.nlevels + level] = C;
    __syncthreads();   // myCand written by this block; make it visible to itself
    __threadfence_block();

    const int N = L.quota;
    // ---- roots (src/ORBextractor.cc:545-586)
    const int nIni = L.nIni;
    for (int i = tid; i < 4 * M; i += 256) childCnt[i] = 0;
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int x = myCand[c] >> 20;
        int r = (int)((float)x / L.hX);
        r = min(r, nIni - 1);
        myNode[c] = (unsigned short)r;
        atomicAdd(&childCnt[r], 1);
    }
    __syncthreads();
    // compact non-empty roots, keeping order
    for (int i = tid; i < M; i += 256) scanA[i] = (i < nIni && childCnt[i] > 0) ? 1 : 0;
    __syncthreads();
    int Lsz = block_excl_scan(scanA, nIni, wsum);
    for (int i = tid; i < nIni; i += 256)
        if (childCnt[i] > 0) {
            const int p = scanA[i];
            NodeRec n;
            n.ulx = (short)(int)(L.hX * (float)i); n.urx = (short)(int)(L.hX * (float)(i + 1));
            n.uly = 0; n.bry = (short)(L.maxBorderY - kMinBorder);
            nodeA[p] = n; cntA[p] = childCnt[i]; seqA[p] = (unsigned short)p;
            scanB[i] = p;   // root -> list position
        }
    __syncthreads();
    for (int c = tid; c < C; c += 256) myNode[c] = (unsigned short)scanB[myNode[c]];
    __syncthreads();

    NodeRec* cur = nodeA; NodeRec* nxt = nodeB;
    int* curCnt = cntA; int* nxtCnt = cntB;
    unsigned short* curSeq = seqA; unsigned short* nxtSeq = seqB;
    bool careful = false, finish = (C == 0);
    int guard = 0;
    while (!finish && guard++ < 64) {
        // 1. children counts of every expandable node
        for (int i = tid; i < 4 * Lsz; i += 256) childCnt[i] = 0;
        __syncthreads();
        // (four candidates per thread and step, their words requested together: a level-0 list is thousands of candidates, and as one dependent load per
        // iteration the two passes of every subdivision round were the kernel's time -- 5.5 ms per step for 0.26 M instructions per image)
        for (int c0 = tid; c0 < C; c0 += 4 * 256) {
            int ni[4];
            uint32_t pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int c = c0 + 256 * u; ni[u] = c < C ? (int)myNode[c] : -1; pp[u] = c < C ? myCand[c] : 0u; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = ni[u];
                if (i >= 0 && curCnt[i] > 1) {
                    int hx, hy;
                    const int q = quadrant(cur[i], (int)(pp[u] >> 20), (int)((pp[u] >> 8) & 0xfff), hx, hy);
                    atomicAdd(&childCnt[4 * i + q], 1);
                }
            }
        }
        __syncthreads();
        // 2. processing order
        int E;
        if (!careful) {
            for (int i = tid; i < Lsz; i += 256) scanA[i] = curCnt[i] > 1 ? 1 : 0;
            __syncthreads();
            E = block_excl_scan(scanA, Lsz, wsum);
            for (int i = tid; i < Lsz; i += 256)
                if (curCnt[i] > 1) procNode[scanA[i]] = (unsigned short)i;
        } else {
            // sort expandable nodes by (count, creation index) descending: bitonic over M keys
            for (int i = tid; i < M; i += 256) {
                uint32_t k = 0;
                if (i < Lsz && curCnt[i] > 1) {
                    k = ((uint32_t)min(curCnt[i], 0xffff) << 16) | curSeq[i];      // (a level holds at most 65535 candidates; creation indices < M <= 32768)
                    seqToNode[curSeq[i]] = (unsigned short)i;
                }
                sortKey[i] = k;
            }
            __syncthreads();
            for (int k = 2; k <= M; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < M; i += 256) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const uint32_t a = sortKey[i], b = sortKey[ixj];
                            const bool desc = (i & k) == 0;
                            if (desc ? (a < b) : (a > b)) { sortKey[i] = b; sortKey[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            if (tid == 0) sh[0] = 0;
            __syncthreads();
            int cntE = 0;
            for (int i = tid; i < M; i += 256)
                if (sortKey[i] != 0) { procNode[i] = seqToNode[sortKey[i] & 0xffff]; ++cntE; }
            atomicAdd(&sh[0], cntE);
            __syncthreads();
            E = sh[0];
        }
        __syncthreads();
        // 3. nc along the processing order, inclusive running size -> cut-off K
        for (int r = tid; r < E; r += 256) {
            const int i = procNode[r];
            scanB[r] = (childCnt[4 * i] > 0) + (childCnt[4 * i + 1] > 0) + (childCnt[4 * i + 2] > 0) + (childCnt[4 * i + 3] > 0);
        }
        __syncthreads();
        block_excl_scan(scanB, E, wsum);   // scanB[r] = children created before rank r
        if (tid == 0) sh[1] = E;
        __syncthreads();
        if (careful) {
            // first rank r whose division brings the list to >= N:  Lsz - (r+1) + excl[r] + nc[r] >= N
            for (int r = tid; r < E; r += 256) {
                const int i = procNode[r];
                const int nc = (childCnt[4 * i] > 0) + (childCnt[4 * i + 1] > 0) + (childCnt[4 * i + 2] > 0) + (childCnt[4 * i + 3] > 0);
                if (Lsz - (r + 1) + scanB[r] + nc >= N) atomicMin(&sh[1], r + 1);
            }
            __syncthreads();
        }
        const int K = sh[1];
        for (int i = tid; i < Lsz; i += 256) rankOf[i] = 0xffff;
        __syncthreads();
        for (int r = tid; r < K; r += 256) rankOf[procNode[r]] = (unsigned short)r;
        if (tid == 0) {
            int T = 0;
            if (K > 0) {
                const int i = procNode[K - 1];
                T = scanB[K - 1] + (childCnt[4 * i] > 0) + (childCnt[4 * i + 1] > 0) + (childCnt[4 * i + 2] > 0) + (childCnt[4 * i + 3] > 0);
            }
            sh[2] = T; sh[3] = 0;
        }
        __syncthreads();
        const int T = sh[2];
        // 4. survivors keep their relative order behind the new children
        for (int i = tid; i < Lsz; i += 256) scanC[i] = rankOf[i] == 0xffff ? 1 : 0;
        __syncthreads();
        block_excl_scan(scanC, Lsz, wsum);
        const int newL = T + (Lsz - K);
        // 5. new node records
        int nExp = 0;
        for (int i = tid; i < Lsz; i += 256) {
            const int r = rankOf[i];
            if (r == 0xffff) {
                const int p = T + scanC[i];
                nxt[p] = cur[i]; nxtCnt[p] = curCnt[i]; nxtSeq[p] = 0;
            } else {
                const NodeRec n = cur[i];
                int hx, hy;
                quadrant(n, 0, 0, hx, hy);
                const int c0 = childCnt[4 * i], c1 = childCnt[4 * i + 1], c2 = childCnt[4 * i + 2], c3 = childCnt[4 * i + 3];
                const int nc = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                const int base = T - (scanB[r] + nc);   // children of later-processed nodes sit in front
                int after = nc, before = 0;
                const int cs[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (cs[q] > 0) {
                        --after;
                        const int p = base + after;
                        NodeRec ch;
                        ch.ulx = (short)((q & 1) ? n.ulx + hx : n.ulx);
                        ch.urx = (short)((q & 1) ? n.urx : n.ulx + hx);
                        ch.uly = (short)((q & 2) ? n.uly + hy : n.uly);
                        ch.bry = (short)((q & 2) ? n.bry : n.uly + hy);
                        nxt[p] = ch; nxtCnt[p] = cs[q];
                        nxtSeq[p] = (unsigned short)(scanB[r] + before);
                        ++before;
                        if (cs[q] > 1) ++nExp;
                        childCnt[4 * i + q] = p + 1;   // reuse: quadrant -> new position (+1), 0 = empty
                    }
                }
            }
        }
        atomicAdd(&sh[3], nExp);
        __syncthreads();
        // 6. move the key points
        for (int c0 = tid; c0 < C; c0 += 4 * 256) {
            int ni[4];
            uint32_t pp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int c = c0 + 256 * u; ni[u] = c < C ? (int)myNode[c] : -1; pp[u] = c < C ? myCand[c] : 0u; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = ni[u], c = c0 + 256 * u;
                if (i < 0) continue;
                if (rankOf[i] == 0xffff) myNode[c] = (unsigned short)(T + scanC[i]);
                else {
                    int hx, hy;
                    const int q = quadrant(cur[i], (int)(pp[u] >> 20), (int)((pp[u] >> 8) & 0xfff), hx, hy);
                    myNode[c] = (unsigned short)(childCnt[4 * i + q] - 1);
                }
            }
        }
        __syncthreads();
        const int nToExpand = sh[3];
        // 7. termination (src/ORBextractor.cc:669-741)
        if (newL >= N || newL == Lsz) finish = true;
        else if (!careful && newL + nToExpand * 3 > N) careful = true;
        Lsz = newL;
        { NodeRec* t = cur; cur = nxt; nxt = t; }
        { int* t = curCnt; curCnt = nxtCnt; nxtCnt = t; }
        { unsigned short* t = curSeq; curSeq = nxtSeq; nxtSeq = t; }
        __syncthreads();
    }
    // ---- best response per leaf, first in candidate order on ties (src/ORBextractor.cc:746-761)
    uint32_t* bestKey = reinterpret_cast<uint32_t*>(childCnt);
    for (int i = tid; i < Lsz; i += 256) bestKey[i] = 0;
    __syncthreads();
    for (int c = tid; c < C; c += 256) atomicMax(&bestKey[myNode[c]], ((myCand[c] & 0xffu) << 16) | (uint32_t)(0xffff - c));
    __syncthreads();
    uint32_t* out = lvlKp + (size_t)img * g.kpTotal + L.kpBase;
    if (Lsz > L.kpCap) { if (tid == 0) atomicOr(status, 2); Lsz = L.kpCap; }
    for (int i = tid; i < Lsz; i += 256) out[i] = myCand[0xffff - (bestKey[i] & 0xffff)];
    if (tid == 0) lvlCount[img * g.nlevels + level] = Lsz;
}

size_t octree_lds_bytes(int M)
{
    return (size_t)M * (2 * sizeof(NodeRec) + 2 * 4 + 4 * 4 + 3 * 4 + 4 + 5 * 2) + 4096 * 4;
}

int launch_orb_octree(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, hipStream_t s)
{
    const size_t lds = octree_lds_bytes(g.maxNodes);
    if (g.maxNodes > 2048) {
        hipLaunchKernelGGL(k_octree<true>, dim3(g.nlevels, n_images), dim3(256), 0, s, b.geom, b.cells, b.cellCount, b.cand, b.candNode,
                           b.candCount, b.lvlKp, b.lvlCount, b.status, b.octSpill, lds);
        OLF_HIP_CHECK(hipGetLastError());
        return OLF_OK;
    }
    // the attribute belongs to the (function, device) pair: set it on whichever device this launch goes to
    if (lds > 64 * 1024)
        OLF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_octree<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    hipLaunchKernelGGL(k_octree<false>, dim3(g.nlevels, n_images), dim3(256), lds, s, b.geom, b.cells, b.cellCount, b.cand, b.candNode,
                       b.candCount, b.lvlKp, b.lvlCount, b.status, nullptr, 0);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
