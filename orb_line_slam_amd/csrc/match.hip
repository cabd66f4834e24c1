// match.hip -- 256-bit Hamming matchers of the path, gfx950.
//   Frame::ComputeStereoMatches            reference src/Frame.cc:702-876
//   matchNNR / match(desc1, desc2, nnr)    reference src/LineMatcher.cpp:42-62, :104-132
//   ORBmatcher::DescriptorDistance         reference src/ORBmatcher.cc:1795-1811 (dense matrix form)
// Bit counting, not contraction: XOR + v_bcnt (popcount) on 64-bit words, one wave per query,
// candidates strided over the 64 lanes and reduced with DPP/shuffle min; no MFMA.
#include "olf_internal.hpp"
#include "device_math.hpp"

namespace olf {

__device__ __forceinline__ int ham256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// ---------------------------------------------------------------------------------------------
// ComputeStereoMatches, one wave per left key point of one stereo pair (images 2p, 2p+1).
constexpr int TH_HIGH = 100, TH_LOW = 50;

// Candidate search of ComputeStereoMatches (src/Frame.cc:719-786), one THREAD per left key point.  The right key
// points stream through LDS in tiles (row span [floor(y-r), ceil(y+r)], octave, x, descriptor) and are read at a
// wave-uniform address; a left key point only pays the Hamming distance for right points that pass the row / octave /
// disparity-range gates.  best = min (distance, iR): the reference scans candidates in increasing iR with a strict '<'.
constexpr int ST_TILE = 128;

// Both key point lists of a pair ordered by image row ((int)y << 16 | index, ascending): a block of the candidate search then covers a
// narrow band of left rows and only has to look at the right key points whose rows can reach it.
__global__ __launch_bounds__(256) void k_stereo_rowsort(const olf_keypoint* __restrict__ kps, const int* __restrict__ counts, int cap, int sortN,
                                                        unsigned* __restrict__ perm)
{
    OLF_SET_GUEST_PRIO();
    extern __shared__ unsigned keys[];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int n = counts[img];
    for (int i = tid; i < sortN; i += 256) {
        unsigned k = 0xffffffffu;
        if (i < n) k = ((unsigned)min(max((int)kps[(size_t)img * cap + i].y, 0), 65534) << 16) | (unsigned)i;
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= sortN; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < sortN; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned a = keys[i], b = keys[ixj];
                    const bool asc = (i & k) == 0;
                    if (asc ? (a > b) : (a < b)) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < n; i += 256) perm[(size_t)img * cap + i] = keys[i];
}

__global__ __launch_bounds__(256) void k_stereo_cand(const OrbGeom* __restrict__ gp, const olf_keypoint* __restrict__ kps,
                                                     const uint8_t* __restrict__ desc, const int* __restrict__ counts, int cap, float mbf,
                                                     float fx, const unsigned* __restrict__ perm, unsigned* __restrict__ bestKey)
{
    OLF_SET_GUEST_PRIO();
    __shared__ uint4 s_d[ST_TILE * 2];
    __shared__ int s_rows[ST_TILE];      // minr | maxr << 16
    __shared__ int s_oct[ST_TILE];
    __shared__ float s_x[ST_TILE];
    __shared__ int s_idx[ST_TILE];
    __shared__ int s_range[2];
    const OrbGeom& g = *gp;
    const int pair = blockIdx.y;
    const int tL = blockIdx.x * 256 + threadIdx.x;           // position in the row-sorted left list
    const int nL = counts[2 * pair], nR = counts[2 * pair + 1];
    if (blockIdx.x * 256 >= nL) return;
    const size_t oL = (size_t)(2 * pair) * cap, oR = (size_t)(2 * pair + 1) * cap;
    const unsigned* pL = perm + oL;
    const unsigned* pR = perm + oR;
    const bool live = tL < nL;
    const int iL = (int)(pL[live ? tL : 0] & 0xffffu);
    olf_keypoint kL = kps[oL + iL];
    const uint4* dLp = reinterpret_cast<const uint4*>(desc + (oL + iL) * OLF_DESC_BYTES);
    const uint4 a0 = dLp[0], a1 = dLp[1];
    const int levelL = kL.octave, row = (int)kL.y;
    const float mb = f_div(mbf, fx);
    const float maxD = f_div(mbf, mb);
    const float minU = f_sub(kL.x, maxD), maxU = kL.x;
    unsigned best = ((unsigned)TH_HIGH << 16) | 0xffffu;
    if (threadIdx.x == 0) {
        // right key points whose band (row +- 2 * scale, rounded outwards) can contain one of this block's rows: a superset, the exact
        // test stays below
        const int rad = (int)ceilf(f_mul(2.0f, g.lv[g.nlevels - 1].scale)) + 2;
        const int first = (int)blockIdx.x * 256, last = min(first + 255, nL - 1);
        const int rlo = (int)(pL[first] >> 16) - rad, rhi = (int)(pL[last] >> 16) + rad;
        int lo = 0, hi = nR;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((int)(pR[m] >> 16) < rlo) lo = m + 1; else hi = m; }
        s_range[0] = lo;
        hi = nR;
        while (lo < hi) { const int m = (lo + hi) >> 1; if ((int)(pR[m] >> 16) <= rhi) lo = m + 1; else hi = m; }
        s_range[1] = lo;
    }
    __syncthreads();
    const int jlo = s_range[0], jhi = s_range[1];
    for (int t0 = jlo; t0 < jhi; t0 += ST_TILE) {
        const int cnt = min(ST_TILE, jhi - t0);
        __syncthreads();
        if (threadIdx.x < cnt) {
            const int iR = (int)(pR[t0 + threadIdx.x] & 0xffffu);
            const olf_keypoint kR = kps[oR + iR];
            const float r = f_mul(2.0f, g.lv[kR.octave].scale);
            const int maxr = (int)ceilf(f_add(kR.y, r)), minr = (int)floorf(f_sub(kR.y, r));
            s_rows[threadIdx.x] = (minr & 0xffff) | (maxr << 16);
            s_oct[threadIdx.x] = kR.octave;
            s_x[threadIdx.x] = kR.x;
            s_idx[threadIdx.x] = iR;
        }
        if (threadIdx.x < 2 * cnt) {
            const int iR = (int)(pR[t0 + (threadIdx.x >> 1)] & 0xffffu);
            s_d[threadIdx.x] = reinterpret_cast<const uint4*>(desc + (oR + iR) * OLF_DESC_BYTES)[(int)(threadIdx.x & 1)];
        }
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const int rw = s_rows[j];
            const int minr = (int)(short)(rw & 0xffff), maxr = rw >> 16;
            if (row < minr || row > maxr) continue;
            const int oc = s_oct[j];
            if (oc < levelL - 1 || oc > levelL + 1) continue;
            const float xr = s_x[j];
            if (!(xr >= minU && xr <= maxU)) continue;
            const unsigned d = (unsigned)ham256(a0, a1, s_d[2 * j], s_d[2 * j + 1]);
            best = min(best, (d << 16) | (unsigned)s_idx[j]);       // min over (distance, iR): the reference scans iR upwards with a strict '<'
        }
    }
    if (live) bestKey[(size_t)pair * cap + iL] = best;
}

__global__ __launch_bounds__(256) void k_stereo_match(const OrbGeom* __restrict__ gp, const uint8_t* __restrict__ pyr,
                                                      const olf_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                      const int* __restrict__ counts, int cap, float mbf, float fx,
                                                      const unsigned* __restrict__ bestKey, float* __restrict__ uRight,
                                                      float* __restrict__ depth, int* __restrict__ sad)
{
    OLF_SET_GUEST_PRIO();
    __shared__ uint8_t s_strip[4][11 * 21 + 1];
    const OrbGeom& g = *gp;
    const int pair = blockIdx.y, lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nL = counts[2 * pair], nR = counts[2 * pair + 1];
    if (iL >= nL) return;
    const size_t oL = (size_t)(2 * pair) * cap, oR = (size_t)(2 * pair + 1) * cap;
    float outU = -1.0f, outD = -1.0f;
    int outS = -1;
    // (what the tests below branch on is requested up front, in two rounds instead of five: the left key point and its best candidate; then the candidate's
    // column and the level's geometry -- for a key point without a candidate from a clamped index, unused)
    const olf_keypoint kL = kps[oL + iL];
    const unsigned best = bestKey[(size_t)pair * cap + iL];
    const int levelL = kL.octave;
    const float vL = kL.y, uL = kL.x;
    asm volatile("" :: "v"(levelL), "v"(vL), "v"(uL), "v"(best));
    const float uR0 = kps[oR + min((int)(best & 0xffffu), max(nR - 1, 0))].x;
    const LevelGeom L = g.lv[min(max(levelL, 0), g.nlevels - 1)];
    asm volatile("" :: "v"(uR0), "v"(L.inv_scale), "v"(L.scale), "v"(L.w), "v"(L.pitch), "v"(L.offset));
    const float mb = f_div(mbf, fx);
    const float maxD = f_div(mbf, mb);
    const float minU = f_sub(uL, maxD), maxU = uL;
    const int row = (int)vL;
    const int bestDist = (int)(best >> 16);
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    if (bestDist < thOrbDist && (best & 0xffffu) != 0xffffu && bestDist < TH_HIGH) {
        const float scaleFactor = L.inv_scale;
        const float scaleduL = roundf(f_mul(kL.x, scaleFactor));
        const float scaledvL = roundf(f_mul(kL.y, scaleFactor));
        const float scaleduR0 = roundf(f_mul(uR0, scaleFactor));
        const int w = 5, Ls = 5;
        const float iniu = scaleduR0 + Ls - w, endu = scaleduR0 + Ls + w + 1;
        if (!(iniu < 0 || endu >= (float)L.w)) {
            const uint8_t* IL = pyr + (size_t)(2 * pair) * g.pyrBytes + L.offset;
            const uint8_t* IR = pyr + (size_t)(2 * pair + 1) * g.pyrBytes + L.offset;
            const int cxL = (int)scaleduL, cy = (int)scaledvL, cxR0 = (int)scaleduR0;
            // the 11 shifted 11x11 windows of the right image overlap in a 21x11 strip: it is staged once in LDS (4 byte loads per lane
            // instead of 33) and every shift reads its window and its centre pixel from there.  All seven loads of a lane -- strip, patch, centre -- are
            // requested before any is used: as a loop that stored each strip byte to LDS before loading the next they were six dependent round trips
            uint8_t* strip = s_strip[threadIdx.x >> 6];
            int sv[4], araw[2], py[2], px[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + 64 * k, ry = idx / 21, rx = idx - ry * 21;
                sv[k] = idx < 11 * 21 ? (int)IR[(size_t)(cy + ry - w) * L.pitch + cxR0 + rx - (w + Ls)] : 0;
            }
            // each lane owns up to two of the 121 patch pixels
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int idx = lane + 64 * k;
                py[k] = idx / 11 - w; px[k] = idx % 11 - w;
                araw[k] = idx < 121 ? (int)IL[(size_t)(cy + py[k]) * L.pitch + cxL + px[k]] : 0;
            }
            const int cL = IL[(size_t)cy * L.pitch + cxL];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int idx = lane + 64 * k; if (idx < 11 * 21) strip[idx] = (uint8_t)sv[k]; }
            int a[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) a[k] = lane + 64 * k < 121 ? araw[k] - cL : 0;
            __builtin_amdgcn_wave_barrier();
            // the 11 sums of absolute differences: per lane two shifts' partial sums share a dword (a sum is below 121 * 255 < 2^15), six wave sums through
            // DPP instead of eleven butterflies through the LDS crossbar (66 dependent ds_bpermute per key point were most of the kernel's latency)
            int part[11];
#pragma unroll
            for (int inc = -Ls; inc <= Ls; ++inc) {
                const int cR = strip[w * 21 + (w + Ls) + inc];
                int s = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = lane + 64 * k;
                    if (idx < 121) {
                        const int b = (int)strip[(py[k] + w) * 21 + px[k] + (w + Ls) + inc] - cR;
                        const int df = a[k] - b;
                        s += df < 0 ? -df : df;
                    }
                }
                part[inc + Ls] = s;
            }
            int dists[11];
#pragma unroll
            for (int k = 0; k < 10; k += 2) {
                const int t = wave_sum_i32(part[k] | (part[k + 1] << 16));
                dists[k] = t & 0xffff; dists[k + 1] = (int)((unsigned)t >> 16);
            }
            dists[10] = wave_sum_i32(part[10]);
            int bestS = 0x7fffffff, bestinc = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k)
                if (dists[k] < bestS) { bestS = dists[k]; bestinc = k - Ls; }
            if (bestinc != -Ls && bestinc != Ls) {
                float dist1 = 0, dist2 = 0, dist3 = 0;
#pragma unroll
                for (int k = 1; k < 10; ++k)
                    if (k - Ls == bestinc) { dist1 = (float)dists[k - 1]; dist2 = (float)dists[k]; dist3 = (float)dists[k + 1]; }
                const float num = f_sub(dist1, dist3);
                const float den = f_mul(2.0f, f_sub(f_add(dist1, dist3), f_mul(2.0f, dist2)));
                const float deltaR = f_div(num, den);
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = f_mul(L.scale, f_add(f_add(scaleduR0, (float)bestinc), deltaR));
                    float disparity = f_sub(uL, bestuR);
                    if (disparity >= 0 && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = 0.01f;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        outD = f_div(mbf, disparity);
                        outU = bestuR;
                        outS = bestS;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        uRight[(size_t)pair * cap + iL] = outU;
        depth[(size_t)pair * cap + iL] = outD;
        sad[(size_t)pair * cap + iL] = outS;
    }
}

// median filter of src/Frame.cc:862-875: drop matches whose SAD >= 1.5*1.4*median(SAD).
__global__ __launch_bounds__(256) void k_stereo_median(const int* __restrict__ counts, int cap, int sortN, float* __restrict__ uRight,
                                                       float* __restrict__ depth, const int* __restrict__ sad)
{
    OLF_SET_GUEST_PRIO();
    extern __shared__ unsigned keys[];
    __shared__ int nMatched;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nL = counts[2 * pair];
    if (tid == 0) nMatched = 0;
    __syncthreads();
    const int* s = sad + (size_t)pair * cap;
    int local = 0;
    for (int i = tid; i < sortN; i += 256) {
        unsigned k = 0xffffffffu;
        if (i < nL && s[i] >= 0) { k = (unsigned)s[i]; ++local; }
        keys[i] = k;
    }
    atomicAdd(&nMatched, local);
    __syncthreads();
    const int m = nMatched;
    if (m == 0) return;   // convention C.5: nothing to filter
    for (int k = 2; k <= sortN; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < sortN; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned a = keys[i], b = keys[ixj];
                    const bool asc = (i & k) == 0;
                    if (asc ? (a > b) : (a < b)) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    const float median = (float)keys[m / 2];
    const float thDist = f_mul(f_mul(1.5f, 1.4f), median);
    for (int i = tid; i < nL; i += 256)
        if (s[i] >= 0 && !((float)s[i] < thDist)) {
            uRight[(size_t)pair * cap + i] = -1.0f;
            depth[(size_t)pair * cap + i] = -1.0f;
        }
}

// ---------------------------------------------------------------------------------------------
// Brute-force 2-nearest-neighbour search (cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2), App. A.10):
// per query the two smallest distances; ties keep the lower train index first.
// Sets are batched: set s has nQ[s] queries at q + s*strideQ*32 and nT[s] train rows at t + s*strideT*32.
//
// All-pairs Hamming is a GEMM: hamming(a, b) = |a| + |b| - 2 a.b over the 256 bits taken as 0/1 vectors, so the distance tile comes out
// of the matrix cores (v_mfma_i32_32x32x32_i8, exact in int32): the train rows are the A operand with bytes {0, 1}, the queries the B
// operand with bytes {0, -2}, and the accumulator starts at |a| + |b|.  A workgroup owns 256 queries (4 waves x two 32-query column
// blocks, their expanded bits held in registers for the whole kernel) and walks the train set in tiles of 32 rows, expanded to bytes
// once per workgroup into LDS (double buffered).  In the C/D layout a lane holds 16 train rows of ONE query (col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), so the running top-2 is per lane: (distance << 16 | train index) keys, three integer
// ops per candidate, and one exchange between the two lane halves at the end.  The k index of an A / B byte only has to be the
// same function of (lane >> 5, byte) on both sides -- any such assignment sums the same 256 products.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int KM_ROWS = 32;          // train rows per tile
constexpr int KM_STRIDE = 272;       // bytes per expanded row in LDS: 256 + 16 (16-byte reads of 16 consecutive rows touch every bank once)
constexpr int KM_PAD_DIST = 0x4000;  // |a| of a row past the end of the train set: never among real candidates, recognised at the end

__device__ __forceinline__ uint32_t spread4(uint32_t nib) { return (nib * 0x00204081u) & 0x01010101u; }   // bits 0..3 -> bytes 0..3

// SHIFT = 12 (train sets of up to 4096 rows, every case on the path): A bytes {0, 64}, B bytes {0, -128}, so a common bit contributes
// -2 * 4096 and an accumulator that starts at ((|a| + |b|) << 12) + train index ends as the finished (distance << 12 | index) key --
// no instruction builds it.  SHIFT = 16 (up to 65535 rows): bytes {0, 1} / {0, -2}, the accumulator is the distance and the key is one
// shift-add.
template <int SHIFT>
__device__ __forceinline__ void knn2_body(uint8_t (*s_a)[KM_ROWS * KM_STRIDE], int (*s_pa)[KM_ROWS], const int set, const int nq, const int nt,
                                          const uint8_t* __restrict__ q, int strideQ, const uint8_t* __restrict__ t, int strideT,
                                          int* __restrict__ idx0, int* __restrict__ dist0, int* __restrict__ dist1)
{
    constexpr bool FUSED = SHIFT == 12;
    constexpr uint32_t A_BYTE = FUSED ? 0x40u : 0x01u, B_BYTE = FUSED ? 0x80u : 0xfeu;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 5, col = lane & 31;

    // B operand: the wave's 2 x 32 queries, 8 k-chunks of 32 bits; this lane carries bits [32 c + 16 g, +16) of query `col`
    v4i bq[2][8];
    int pb[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int iq = blockIdx.x * 256 + wv * 64 + cb * 32 + col;
        uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
        if (iq < nq) {
            const uint4* qp = reinterpret_cast<const uint4*>(q + ((size_t)set * strideQ + iq) * OLF_DESC_BYTES);
            d0 = qp[0]; d1 = qp[1];
        }
        const uint32_t d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        int p = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            p += __popc(d[c]);
            const uint32_t h = (d[c] >> (16 * g)) & 0xffffu;
            v4i f;
            f.x = (int)(spread4(h & 15u) * B_BYTE); f.y = (int)(spread4((h >> 4) & 15u) * B_BYTE);
            f.z = (int)(spread4((h >> 8) & 15u) * B_BYTE); f.w = (int)(spread4(h >> 12) * B_BYTE);
            bq[cb][c] = f;
        }
        pb[cb] = FUSED ? p << 12 : p;
    }

    // A operand staging: thread (row r = tid / 8, word w = tid % 8) expands 32 bits of train row t0 + r into 32 bytes
    const int sr = threadIdx.x >> 3, sw = threadIdx.x & 7;
    const uint32_t* tp = reinterpret_cast<const uint32_t*>(t + (size_t)set * strideT * OLF_DESC_BYTES);
    auto stage = [&](int buf, int t0) {
        const bool in = t0 + sr < nt;
        const uint32_t w = in ? tp[(size_t)(t0 + sr) * 8 + sw] : 0u;
        uint4 lo, hi;
        lo.x = spread4(w & 15u) * A_BYTE; lo.y = spread4((w >> 4) & 15u) * A_BYTE; lo.z = spread4((w >> 8) & 15u) * A_BYTE;
        lo.w = spread4((w >> 12) & 15u) * A_BYTE;
        hi.x = spread4((w >> 16) & 15u) * A_BYTE; hi.y = spread4((w >> 20) & 15u) * A_BYTE; hi.z = spread4((w >> 24) & 15u) * A_BYTE;
        hi.w = spread4(w >> 28) * A_BYTE;
        uint4* dst = reinterpret_cast<uint4*>(&s_a[buf][sr * KM_STRIDE + sw * 32]);
        dst[0] = lo; dst[1] = hi;
        int p = __popc(w);                                   // |a| of the row: sum over its 8 words (8 adjacent lanes)
        p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4);
        if (!in) p = KM_PAD_DIST;
        if (sw == 0) s_pa[buf][sr] = FUSED ? (p << 12) + t0 + sr : p;
    };

    unsigned k0[2] = {0xffffffffu, 0xffffffffu}, k1[2] = {0xffffffffu, 0xffffffffu};
    if (nt > 0) stage(0, 0);
    __syncthreads();
    int buf = 0;
    for (int t0 = 0; t0 < nt; t0 += KM_ROWS) {
        if (t0 + KM_ROWS < nt) stage(buf ^ 1, t0 + KM_ROWS);
        // accumulators start at |a| + |b| (FUSED: shifted, plus the row's index)
        v16i acc0, acc1;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int4 pa = *reinterpret_cast<const int4*>(&s_pa[buf][8 * q4 + 4 * g]);
            acc0[4 * q4 + 0] = pa.x + pb[0]; acc0[4 * q4 + 1] = pa.y + pb[0]; acc0[4 * q4 + 2] = pa.z + pb[0]; acc0[4 * q4 + 3] = pa.w + pb[0];
            acc1[4 * q4 + 0] = pa.x + pb[1]; acc1[4 * q4 + 1] = pa.y + pb[1]; acc1[4 * q4 + 2] = pa.z + pb[1]; acc1[4 * q4 + 3] = pa.w + pb[1];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const v4i a = *reinterpret_cast<const v4i*>(&s_a[buf][col * KM_STRIDE + c * 32 + g * 16]);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][c], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][c], acc1, 0, 0, 0);
        }
        const unsigned rowbase = (unsigned)(t0 + 4 * g);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned ka, kb;
            if (FUSED) { ka = (unsigned)acc0[r]; kb = (unsigned)acc1[r]; }
            else {
                const unsigned idx = rowbase + (unsigned)((r & 3) + 8 * (r >> 2));
                ka = ((unsigned)acc0[r] << 16) + idx; kb = ((unsigned)acc1[r] << 16) + idx;
            }
            k1[0] = min(k1[0], max(k0[0], ka)); k0[0] = min(k0[0], ka);
            k1[1] = min(k1[1], max(k0[1], kb)); k0[1] = min(k0[1], kb);
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        // the other half of the rows of this query lives in lane ^ 32
        const unsigned o0 = (unsigned)__shfl_xor((int)k0[cb], 32), o1 = (unsigned)__shfl_xor((int)k1[cb], 32);
        unsigned m0 = min(k0[cb], o0), m1 = min(max(k0[cb], o0), min(k1[cb], o1));
        if (m0 >= ((unsigned)KM_PAD_DIST << SHIFT)) m0 = 0xffffffffu;          // padding rows are not candidates
        if (m1 >= ((unsigned)KM_PAD_DIST << SHIFT)) m1 = 0xffffffffu;
        const int iq = blockIdx.x * 256 + wv * 64 + cb * 32 + col;
        if (g == 0 && iq < nq) {
            const size_t o = (size_t)set * strideQ + iq;
            idx0[o] = m0 == 0xffffffffu ? -1 : (int)(m0 & ((1u << SHIFT) - 1u));
            dist0[o] = m0 == 0xffffffffu ? 0x7fffffff : (int)(m0 >> SHIFT);
            dist1[o] = m1 == 0xffffffffu ? 0x7fffffff : (int)(m1 >> SHIFT);
        }
    }
}

__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, const int* __restrict__ nQ, int strideQ, int qSetStep,
                                              const uint8_t* __restrict__ t, const int* __restrict__ nT, int strideT, int tSetStep,
                                              int* __restrict__ idx0, int* __restrict__ dist0, int* __restrict__ dist1)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_a[2][KM_ROWS * KM_STRIDE];
    __shared__ __attribute__((aligned(16))) int s_pa[2][KM_ROWS];
    const int set = blockIdx.y;
    const int nq = nQ[set * qSetStep], nt = nT[set * tSetStep];
    if (blockIdx.x * 256 >= nq) return;                      // whole block beyond the query set
    if (nt <= 4096) knn2_body<12>(s_a, s_pa, set, nq, nt, q, strideQ, t, strideT, idx0, dist0, dist1);
    else knn2_body<16>(s_a, s_pa, set, nq, nt, q, strideQ, t, strideT, idx0, dist0, dist1);
}

static void launch_knn2_kernel(dim3 grid, hipStream_t s, const uint8_t* q, const int* nQ, int strideQ, int qStep, const uint8_t* t, const int* nT,
                               int strideT, int tStep, int* idx0, int* dist0, int* dist1)
{
    hipLaunchKernelGGL(k_knn2, grid, dim3(256), 0, s, q, nQ, strideQ, qStep, t, nT, strideT, tStep, idx0, dist0, dist1);
}

// ratio test of matchNNR (src/LineMatcher.cpp:54-59) + mutual check of match() (:121-127)
__global__ __launch_bounds__(256) void k_ratio_mutual(const int* __restrict__ nA, int strideA, int aStep, const int* __restrict__ nB,
                                                      int strideB, int bStep, const int* __restrict__ idxAB,
                                                      const int* __restrict__ d0AB, const int* __restrict__ d1AB,
                                                      const int* __restrict__ idxBA, const int* __restrict__ d0BA,
                                                      const int* __restrict__ d1BA, float nnr, int best_lr, int* __restrict__ m12)
{
    const int set = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int na = nA[set * aStep], nb = nB[set * bStep];
    if (i >= na) return;
    const size_t oa = (size_t)set * strideA + i;
    int m = -1;
    if (nb >= 2 && (float)d0AB[oa] < f_mul((float)d1AB[oa], nnr)) m = idxAB[oa];
    if (m >= 0 && best_lr) {
        const size_t ob = (size_t)set * strideB + m;
        int back = -1;
        if (na >= 2 && (float)d0BA[ob] < f_mul((float)d1BA[ob], nnr)) back = idxBA[ob];
        if (back != i) m = -1;
    }
    m12[oa] = m;
}

// Candidate-list Hamming (the distance part of ORBmatcher::SearchByProjection / SearchByBoW, src/ORBmatcher.cc:1395-1431,
// :199-234): CSR lists of train indices per query, one distance per (query, candidate) pair in list order.  The greedy,
// order-dependent resolution (already-matched skips, MapPoint state) stays on the host, SURVEY 8(b) / App. C.7.
__global__ __launch_bounds__(256) void k_match_candidates(const uint8_t* __restrict__ q, int nQ, const uint8_t* __restrict__ t, int nT,
                                                          const int* __restrict__ offs, const int* __restrict__ cand, uint16_t* __restrict__ out)
{
    const int iq = blockIdx.x * 256 + threadIdx.x;
    if (iq >= nQ) return;
    const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)iq * OLF_DESC_BYTES);
    const uint4 a0 = qp[0], a1 = qp[1];
    for (int k = offs[iq]; k < offs[iq + 1]; ++k) {
        const int j = cand[k];
        uint16_t d = 0xffff;
        if (j >= 0 && j < nT) {
            const uint4* tp = reinterpret_cast<const uint4*>(t + (size_t)j * OLF_DESC_BYTES);
            d = (uint16_t)ham256(a0, a1, tp[0], tp[1]);
        }
        out[k] = d;
    }
}

int launch_match_candidates(const uint8_t* q, int nQ, const uint8_t* t, int nT, const int* offs, const int* cand, uint16_t* out, hipStream_t s)
{
    if (nQ <= 0) return OLF_OK;
    hipLaunchKernelGGL(k_match_candidates, dim3((nQ + 255) / 256), dim3(256), 0, s, q, nQ, t, nT, offs, cand, out);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

// dense distance matrix (DescriptorDistance over all pairs), int16 out[nA][nB]
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t* __restrict__ a, int nA, const uint8_t* __restrict__ b, int nB,
                                                        uint16_t* __restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= nB || i >= nA) return;
    const uint4* ap = reinterpret_cast<const uint4*>(a + (size_t)i * OLF_DESC_BYTES);
    const uint4* bp = reinterpret_cast<const uint4*>(b + (size_t)j * OLF_DESC_BYTES);
    out[(size_t)i * nB + j] = (uint16_t)ham256(ap[0], ap[1], bp[0], bp[1]);
}

// MapPoint / MapLine::ComputeDistinctiveDescriptors (src/MapPoint.cc:254-318, src/MapLine.cc:257-322): among the N descriptors
// observing one landmark, the one whose median Hamming distance to all N (itself included, distance 0) is smallest; median = element
// floor(0.5 * (N - 1)) of the sorted row; the first minimum wins.  One 64-thread workgroup per landmark, descriptors staged in LDS;
// a row's median is found by bisection on the distance value (count(d <= v) >= rank + 1), so no N x N matrix is ever stored.
constexpr int kDistinctMax = 1024;
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ desc, const int* __restrict__ offs, int* __restrict__ best)
{
    __shared__ uint4 s_d[2 * kDistinctMax];
    const int p = blockIdx.x, b = offs[p], N = offs[p + 1] - b, lane = threadIdx.x;
    if (N <= 0) { if (lane == 0) best[p] = -1; return; }
    const uint4* src = reinterpret_cast<const uint4*>(desc + (size_t)b * OLF_DESC_BYTES);
    for (int i = lane; i < 2 * N; i += 64) s_d[i] = src[i];
    __syncthreads();
    const int rank = (int)(0.5 * (double)(N - 1));
    unsigned key = 0xffffffffu;
    for (int i = lane; i < N; i += 64) {
        const uint4 a0 = s_d[2 * i], a1 = s_d[2 * i + 1];
        int lo = 0, hi = 256;                     // smallest v with count(d <= v) > rank
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; ++j) cnt += ham256(a0, a1, s_d[2 * j], s_d[2 * j + 1]) <= mid;
            if (cnt > rank) hi = mid; else lo = mid + 1;
        }
        key = min(key, ((unsigned)lo << 16) | (unsigned)i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
    if (lane == 0) best[p] = (int)(key & 0xffffu);
}

int launch_distinctive(const uint8_t* desc, const int* offs, int n_points, int* best, hipStream_t s)
{
    if (n_points <= 0) return OLF_OK;
    hipLaunchKernelGGL(k_distinctive, dim3(n_points), dim3(64), 0, s, desc, offs, best);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

// ---------------------------------------------------------------------------------------------
int launch_stereo_points(const OrbGeom& g, const OrbDeviceBufs& b, int n_pairs, const olf_keypoint* d_kps, const uint8_t* d_desc,
                         const int* d_counts, int cap, float mbf, float fx, float* d_uRight, float* d_depth, int* d_sad, int* d_bestKey,
                         unsigned* d_perm /* [2 * n_pairs][cap] */, hipStream_t s)
{
    if (cap > 65534) { set_error("stereo points: more than 65534 key points per image"); return OLF_ERR_CAPACITY; }   // 16-bit indices in the keys
    // d_sad doubles as the (distance, index) scratch of the candidate search until k_stereo_match overwrites it per key point
    unsigned* bestKey = reinterpret_cast<unsigned*>(d_bestKey);
    int sortK = 64;
    while (sortK < cap) sortK <<= 1;
    hipLaunchKernelGGL(k_stereo_rowsort, dim3(2 * n_pairs), dim3(256), sortK * sizeof(unsigned), s, d_kps, d_counts, cap, sortK, d_perm);
    hipLaunchKernelGGL(k_stereo_cand, dim3((cap + 255) / 256, n_pairs), dim3(256), 0, s, b.geom, d_kps, d_desc, d_counts, cap, mbf, fx, d_perm,
                       bestKey);
    hipLaunchKernelGGL(k_stereo_match, dim3((cap + 3) / 4, n_pairs), dim3(256), 0, s, b.geom, b.pyr, d_kps, d_desc, d_counts, cap, mbf,
                       fx, bestKey, d_uRight, d_depth, d_sad);
    int sortN = 64;
    while (sortN < cap) sortN <<= 1;
    hipLaunchKernelGGL(k_stereo_median, dim3(n_pairs), dim3(256), sortN * sizeof(unsigned), s, d_counts, cap, sortN, d_uRight, d_depth,
                       d_sad);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_match_bf(const uint8_t* dA, const int* nA, int strideA, int aStep, const uint8_t* dB, const int* nB, int strideB, int bStep,
                    int n_sets, float nnr, int best_lr, int* ws /* 3*(strideA+strideB)*n_sets ints */, int* m12, hipStream_t s)
{
    if (strideA > 65535 || strideB > 65535) { set_error("match: more than 65535 descriptors per set"); return OLF_ERR_CAPACITY; }   // 16-bit index in the keys
    int* idxAB = ws; int* d0AB = idxAB + (size_t)n_sets * strideA; int* d1AB = d0AB + (size_t)n_sets * strideA;
    int* idxBA = d1AB + (size_t)n_sets * strideA; int* d0BA = idxBA + (size_t)n_sets * strideB; int* d1BA = d0BA + (size_t)n_sets * strideB;
    launch_knn2_kernel(dim3((strideA + 255) / 256, n_sets), s, dA, nA, strideA, aStep, dB, nB, strideB, bStep, idxAB, d0AB, d1AB);
    if (best_lr)
        launch_knn2_kernel(dim3((strideB + 255) / 256, n_sets), s, dB, nB, strideB, bStep, dA, nA, strideA, aStep, idxBA, d0BA, d1BA);
    hipLaunchKernelGGL(k_ratio_mutual, dim3((strideA + 255) / 256, n_sets), dim3(256), 0, s, nA, strideA, aStep, nB, strideB, bStep, idxAB, d0AB,
                       d1AB, idxBA, d0BA, d1BA, nnr, best_lr, m12);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_knn2(const uint8_t* dA, const int* nA, int strideA, const uint8_t* dB, const int* nB, int strideB, int n_sets, int* idx0,
                int* dist0, int* dist1, hipStream_t s)
{
    if (strideB > 65535) { set_error("knn2: more than 65535 train descriptors per set"); return OLF_ERR_CAPACITY; }
    launch_knn2_kernel(dim3((strideA + 255) / 256, n_sets), s, dA, nA, strideA, 1, dB, nB, strideB, 1, idx0, dist0, dist1);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_hamming_matrix(const uint8_t* a, int nA, const uint8_t* b, int nB, uint16_t* out, hipStream_t s)
{
    if (nA <= 0 || nB <= 0) return OLF_OK;
    hipLaunchKernelGGL(k_hamming_matrix, dim3((nB + 255) / 256, nA), dim3(256), 0, s, a, nA, b, nB, out);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
