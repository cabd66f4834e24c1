// orb_pyramid.hip -- gfx950 kernels for the image side of ORBextractor::operator()
// (reference src/ORBextractor.cc): ComputePyramid (:1109-1134), the FAST-9/16 score map and the
// per-cell dual-threshold detection of ComputeKeyPointsOctTree (:791-831), and the 7x7 sigma-2
// GaussianBlur of the descriptor stage (:1087-1088).
//
// All of it is u8/int32 stencil work bounded by HBM bandwidth: each pass reads its level once
// (coalesced rows, 64-byte aligned pitch), stages a halo tile in LDS and writes its output once.
// No MFMA: there is no contraction anywhere on this path.
#include "olf_internal.hpp"

namespace olf {

// ---------------------------------------------------------------------------------------------
// ingest: caller images [n][H][in_pitch] -> level 0 of the pyramid block (aligned pitch)
__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ in, uint8_t* __restrict__ pyr,
                                                int W, int H, int in_pitch, int pitch0, int pyrBytes)
{
    // 16 pixels per thread: one (possibly unaligned) 16-byte load, one aligned 16-byte store; the level pitch is a multiple of 64, so the
    // store may run past W into the row padding, the load may not run past the caller's row
    const int img = blockIdx.y;
    const int segs = (W + 15) >> 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= segs * H) return;
    const int y = idx / segs, x = (idx - y * segs) * 16;
    const uint8_t* src = in + (size_t)img * H * in_pitch + (size_t)y * in_pitch + x;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (x + 16 <= W) __builtin_memcpy(&v, src, 16);
    else {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (x + k < W) w[k >> 2] |= (uint32_t)src[k] << (8 * (k & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4*>(pyr + (size_t)img * pyrBytes + (size_t)y * pitch0 + x) = v;
}

// ---------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR) 8UC1, level l from level l-1 (App. A.2): 11-bit coefficients from the
// host tables, horizontal int32 pass, vertical ((b*(h>>4))>>16) pass, +2 >> 2.
__global__ __launch_bounds__(256) void k_resize(uint8_t* __restrict__ pyr, int pyrBytes, LevelGeom P, LevelGeom L,
                                                const ResizeCoef* __restrict__ rx, const ResizeCoef* __restrict__ ry)
{
    const int img = blockIdx.y;
    const int quads = (L.w + 3) >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= quads * L.h) return;
    const int dy = idx / quads, dx0 = (idx - dy * quads) * 4;
    const uint8_t* src = pyr + (size_t)img * pyrBytes + P.offset;
    const ResizeCoef cy = ry[L.resizeTabY + dy];
    const int y0 = min(max((int)cy.ofs, 0), P.h - 1), y1 = min(max((int)cy.ofs + 1, 0), P.h - 1);
    const uint8_t* S0 = src + (size_t)y0 * P.pitch;
    const uint8_t* S1 = src + (size_t)y1 * P.pitch;
    const int b0 = cy.a0, b1 = cy.a1;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dx = dx0 + k;
        if (dx < L.w) {
            const ResizeCoef cx = rx[L.resizeTabX + dx];
            const int sx = cx.ofs, sx1 = min(sx + 1, P.w - 1);
            const int h0 = S0[sx] * cx.a0 + S0[sx1] * cx.a1;
            const int h1 = S1[sx] * cx.a0 + S1[sx1] * cx.a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xff) << (8 * k);
        }
    }
    *reinterpret_cast<uint32_t*>(pyr + (size_t)img * pyrBytes + L.offset + (size_t)dy * L.pitch + dx0) = out;
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16 score map (App. A.4).  score(p) = max over the sixteen 9-arcs of min|I(p)-I(ring)| - 1,
// which is what cv::FAST's cornerScore returns for any threshold at which p is a corner; p is a
// corner at threshold t  <=>  score >= t.  We store score where score >= minThFAST, else 0.
constexpr int FT_W = 128;                            // processed tile: 128 x FT_H pixels, FT_H = NT / 8 (32 rows for 256 threads); LDS holds bytes x0-4 .. x0+131 of rows y0-3 .. y0+FT_H+2
constexpr int FT_EW = 124;                           // emitted part: columns 2 .. 125, rows 1 .. FT_H - 2 (the rest is the NMS halo of the neighbours)
constexpr int FT_INW = (FT_W + 8) / 4;


// FAST-9/16 corners of one level, straight into the per-cell candidate lists of ComputeKeyPointsOctTree (src/ORBextractor.cc:791-831):
// cv::FAST(cell sub-image, threshold, nonmaxSuppression = true) keeps the strict 3x3 local maxima of the corner score inside the cell's
// interior (the sub-image minus its 3-pixel frame; everything outside counts as 0).  Tiles overlap by the 1-pixel NMS halo, so a tile
// scores its corners into an LDS score tile, suppresses non-maxima there (neighbours in another cell do not count) and appends the
// survivors to their cell's list.  No dense score map exists in memory; k_cells_sort orders each list and applies the dual threshold.
// Footprint: 17.4 KB of LDS and <= 64 VGPRs, so that TWO blocks per CU fit beside the 24 resident growth agents (they leave 37 KB of LDS, two wave
// slots and 128 VGPRs per SIMD): in the fused entry this kernel runs in the agents' shadow (api.cpp, OLF_SCHED).
// NT threads per block: 256 (128 x 32 tile, 17.4 KB of LDS) or 128 (128 x 16, 9 KB): beside the resident growth agents a CU has eight free wave slots and
// 40 KB of LDS -- two big blocks or four small ones; the small ones overlap each other's barriers and loads better (OLF_FAST_NT, api.cpp schedule)
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_fast_score(const uint8_t* __restrict__ pyr, int pyrBytes, LevelGeom L, int minTh,
                  uint32_t* __restrict__ cells, int* __restrict__ cellCount, int totalCells, int cellCap)
{
    OLF_SET_GUEST_PRIO();
    constexpr int FT_H = NT / 8, FT_EH = FT_H - 2, FT_INH = FT_H + 6;
    __shared__ uint32_t tile[FT_INH * FT_INW];
    __shared__ unsigned short s_cand[FT_W * FT_H];      // A1's survivors, compacted in place to the corners by A2 (s_list)
    unsigned short* const s_list = s_cand;
    __shared__ uint8_t s_score[FT_H * FT_W];
    __shared__ int s_nc, s_n;
    const int img = blockIdx.z;
    // scores exist on x in [minBorder+3, maxBorderX-3), y likewise; tile origins are 4-byte aligned
    const int x0 = kMinBorder - 4 + blockIdx.x * FT_EW, y0 = kMinBorder + 2 + blockIdx.y * FT_EH;
    const int xBeg = kMinBorder + 3, yBeg = kMinBorder + 3, xEnd = L.maxBorderX - 3, yEnd = L.maxBorderY - 3;
    const uint8_t* src = pyr + (size_t)img * pyrBytes + L.offset;
    {   // (all of a thread's tile words are requested before the first is stored: as a loop of load -> LDS store the block started with six dependent round trips)
        constexpr int NLD = (FT_INH * FT_INW + NT - 1) / NT;
        uint32_t tv[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = threadIdx.x + k * NT;
            const int r = i / FT_INW, j = i - r * FT_INW;
            const int gy = min(y0 - 3 + r, L.h - 1);
            const int xw = min(x0 - 4 + 4 * j, L.pitch - 4);          // rows are 64-byte aligned and padded to the pitch
            tv[k] = i < FT_INH * FT_INW ? *reinterpret_cast<const uint32_t*>(src + (size_t)gy * L.pitch + xw) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int i = threadIdx.x + k * NT; if (i < FT_INH * FT_INW) tile[i] = tv[k]; }
    }
    reinterpret_cast<uint4*>(s_score)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);      // (FT_H * FT_W bytes = NT x 16)
    if (threadIdx.x == 0) { s_nc = 0; s_n = 0; }
    __syncthreads();
    const int q = threadIdx.x & 31, ry0 = (threadIdx.x >> 5) * 4;
    // ---- phase A1: high-speed rejection on the thread's 4x4 pixels (tile rows ry0 .. ry0+9, tile bytes 4q .. 4q+11; pixel p sits at
    // byte 4q+4+p).  A 9-arc of the 16-ring always covers two ADJACENT compass points (N/E/S/W), so a pixel can only be a corner if two
    // adjacent compass points are both brighter than v+t or both darker than v-t.  Survivors are queued (one LDS atomic per thread that has any).
    // SWAR: the four pixels of a row are the middle dword of (w0, w1, w2); a v_perm_b32 puts two of them (even / odd) into the 16-bit halves of a word, the
    // compass neighbours likewise ((x, y +- 3): the same bytes of the rows three above / below; (x +- 3, y): bytes cut out of two dwords by the same
    // instruction).  With the bias K = 512 - (t + 1) per half, n + (K - v) has bit 9 set iff n > v + t and (v + K) - n iff n < v - t; neither can borrow from
    // or carry into the other half (1 <= field <= 765), so one 32-bit add / sub tests two pixels against one compass point.
    uint32_t w[10][3];
#pragma unroll
    for (int r = 0; r < 10; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) w[r][k] = tile[(ry0 + r) * FT_INW + q + k];
    uint32_t hE[10], hO[10];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        hE[r] = __builtin_amdgcn_perm(w[r][1], w[r][1], 0x0c020c00u);      // pixels 0, 2
        hO[r] = __builtin_amdgcn_perm(w[r][1], w[r][1], 0x0c030c01u);      // pixels 1, 3
    }
    const uint32_t K = 0x02000200u - (uint32_t)(minTh + 1) * 0x00010001u, C9 = 0x02000200u;
    uint32_t acc = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int c = rr + 3;
        const uint32_t eE = __builtin_amdgcn_perm(w[c][2], w[c][1], 0x0c050c03u), eO = __builtin_amdgcn_perm(w[c][2], w[c][1], 0x0c060c04u);   // x + 3: bytes 7, 9 / 8, 10
        const uint32_t wE = __builtin_amdgcn_perm(w[c][1], w[c][0], 0x0c030c01u), wO = __builtin_amdgcn_perm(w[c][1], w[c][0], 0x0c040c02u);   // x - 3: bytes 1, 3 / 2, 4
        uint32_t ps[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t V = h ? hO[c] : hE[c], N8 = h ? hO[c - 3] : hE[c - 3], N0 = h ? hO[c + 3] : hE[c + 3], N4 = h ? eO : eE, N12 = h ? wO : wE;
            const uint32_t Bb = K - V, Bd = V + K;
            const uint32_t br = ((N0 + Bb) | (N8 + Bb)) & ((N4 + Bb) | (N12 + Bb));
            const uint32_t dk = ((Bd - N0) | (Bd - N8)) & ((Bd - N4) | (Bd - N12));
            ps[h] = br | dk;
        }
        acc = (acc >> 2) | (ps[0] & C9) | ((ps[1] << 1) & (C9 << 1));
    }
    acc >>= 3;      // pixel p of row rr: bit 2 rr + (p & 1) + 16 (p >> 1)
    {   // scores exist on [xBeg, xEnd) x [yBeg, yEnd) only
        uint32_t cols = 0, bm = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const int gx = x0 + 4 * q + p; if (gx >= xBeg && gx < xEnd) cols |= 1u << ((p & 1) + 16 * (p >> 1)); }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { const int gy = y0 + ry0 + rr; if (gy >= yBeg && gy < yEnd) bm |= cols << (2 * rr); }
        acc &= bm;
    }
    if (acc) {      // survivors are queued (the order inside the queue is immaterial: phase A2 compacts it again and k_cells_sort orders the result)
        int pos = atomicAdd(&s_nc, __popc(acc));
        while (acc) {
            const int b = __builtin_ctz(acc);
            acc &= acc - 1u;
            const int rr = (b & 15) >> 1, p = (b & 1) + 2 * (b >> 4);
            s_cand[pos++] = (unsigned short)((ry0 + rr) * FT_W + 4 * q + p);
        }
    }
    __syncthreads();
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
    constexpr int P = FT_INW * 4;
    // ---- phase A2 + B (round 4): the corner score of every survivor, one per lane (dense), in packed 16-bit arithmetic.  score = max over the sixteen 9-arcs of
    // the arc's minimum of d (ring darker) or of -d (ring brighter), minus 1 -- cv::FAST's cornerScore -- and the survivor IS a corner at minThFAST iff that score
    // reaches it, so the brighter / darker masks and the contiguity test of rounds 1-3 (190 instructions per survivor, then 182 more per corner for the score)
    // are one computation: register i holds (d[i], d[i + 8]) as two int16, a rotation of the ring by one position is "next register" (the last one wraps into
    // the first with its halves swapped), and min over 2, 4, 8 and 9 consecutive positions is four rounds of v_pk_min_i16.
    // (in place: a round's 256 entries are all read before any corner is appended, and the append position never passes the round's end)
    typedef short s2v __attribute__((ext_vector_type(2)));
    const int nCand = s_nc;
    for (int i0 = 0; i0 < nCand; i0 += NT) {
        const int i = i0 + threadIdx.x;
        int sc = 0, id = 0;
        if (i < nCand) {
            id = s_cand[i];
            const int ty = id / FT_W, tx = id - ty * FT_W;
            const uint8_t* c = tb + (ty + 3) * P + tx + 4;
            const int v = c[0];
            int ring[16];
            ring[0] = c[3 * P];   ring[1] = c[3 * P + 1];   ring[2] = c[2 * P + 2];   ring[3] = c[P + 3];
            ring[4] = c[3];       ring[5] = c[-P + 3];      ring[6] = c[-2 * P + 2];  ring[7] = c[-3 * P + 1];
            ring[8] = c[-3 * P];  ring[9] = c[-3 * P - 1];  ring[10] = c[-2 * P - 2]; ring[11] = c[-P - 3];
            ring[12] = c[-3];     ring[13] = c[P - 3];      ring[14] = c[2 * P - 2];  ring[15] = c[3 * P - 1];
            const s2v vv = {(short)v, (short)v};
            s2v R[8], a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const s2v rr = {(short)ring[k], (short)ring[k + 8]};
                R[k] = vv - rr;
            }
            auto sw = [](s2v x) -> s2v { return __builtin_shufflevector(x, x, 1, 0); };
            // darker ring: max over the arcs of the arc's minimum of d.  a = min over 2 consecutive positions, b = over 4, a = over 8; nine = position k with the
            // eight that follow it (one half-swap per round: only the last register wraps into the first)
            s2v bestD = {(short)-1000, (short)-1000};
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_min(R[k], k < 7 ? R[k + 1] : sw(R[0]));
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = __builtin_elementwise_min(a[k], k < 6 ? a[k + 2] : sw(a[k - 6]));
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_min(b[k], k < 4 ? b[k + 4] : sw(b[k - 4]));
            const s2v a0s = sw(a[0]);
#pragma unroll
            for (int k = 0; k < 8; ++k) bestD = __builtin_elementwise_max(bestD, __builtin_elementwise_min(R[k], k < 7 ? a[k + 1] : a0s));
            // brighter ring: max over the arcs of min(-d) = -(min over the arcs of max(d)): the same chain with the two operations exchanged
            s2v bestB = {(short)1000, (short)1000};
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_max(R[k], k < 7 ? R[k + 1] : sw(R[0]));
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = __builtin_elementwise_max(a[k], k < 6 ? a[k + 2] : sw(a[k - 6]));
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_max(b[k], k < 4 ? b[k + 4] : sw(b[k - 4]));
            const s2v a0b = sw(a[0]);
#pragma unroll
            for (int k = 0; k < 8; ++k) bestB = __builtin_elementwise_min(bestB, __builtin_elementwise_max(R[k], k < 7 ? a[k + 1] : a0b));
            const s2v best = __builtin_elementwise_max(bestD, -bestB);
            sc = max((int)best.x, (int)best.y) - 1;
        }
        __syncthreads();
        if (sc >= minTh) {
            s_score[id] = (uint8_t)sc;
            s_list[atomicAdd(&s_n, 1)] = (unsigned short)id;
        }
    }
    __syncthreads();
    const int nC = s_n;
    // ---- phase C: strict 3x3 non-maximum suppression inside the cell interior, survivors appended to their cell
    for (int i = threadIdx.x; i < nC; i += NT) {
        const int id = s_list[i], ty = id / FT_W, tx = id - ty * FT_W;
        if (tx < 2 || tx >= 2 + FT_EW || ty < 1 || ty >= 1 + FT_EH) continue;           // halo: emitted by the neighbouring tile
        const int cur = s_score[id];
        if (cur == 0) continue;
        const int gx = x0 + tx, gy = y0 + ty;
        // (the cell of the corner through the host's multiply-shift pair: two 30-instruction divisions per corner otherwise)
        const int cj = L.wCellM ? (int)(((uint32_t)(gx - xBeg) * L.wCellM) >> 20) : (gx - xBeg) / L.wCell;
        const int ci = L.hCellM ? (int)(((uint32_t)(gy - yBeg) * L.hCellM) >> 20) : (gy - yBeg) / L.hCell;
        const int lx = (gx - xBeg) - cj * L.wCell, ly = (gy - yBeg) - ci * L.hCell;
        // neighbours outside the cell interior (or outside the scored area, where the tile holds 0) do not count
        const bool xl = lx > 0, xr = lx < L.wCell - 1 && gx + 1 < xEnd, yu = ly > 0, yd = ly < L.hCell - 1 && gy + 1 < yEnd;
        const uint8_t* sp = s_score + id;
        int nb = 0;
        if (yu) { nb = max(nb, (int)sp[-FT_W]); if (xl) nb = max(nb, (int)sp[-FT_W - 1]); if (xr) nb = max(nb, (int)sp[-FT_W + 1]); }
        if (yd) { nb = max(nb, (int)sp[FT_W]); if (xl) nb = max(nb, (int)sp[FT_W - 1]); if (xr) nb = max(nb, (int)sp[FT_W + 1]); }
        if (xl) nb = max(nb, (int)sp[-1]);
        if (xr) nb = max(nb, (int)sp[1]);
        if (cur > nb) {
            const int cell = L.cellBase + ci * L.nCols + cj;
            const int pos = atomicAdd(&cellCount[(size_t)img * totalCells + cell], 1);
            if (pos < cellCap)      // (cannot overflow: strict local maxima are at most one per 2x2 block, cellCap is that bound)
                cells[((size_t)img * totalCells + cell) * cellCap + pos] =
                    ((uint32_t)(gy - kMinBorder) << 20) | ((uint32_t)(gx - kMinBorder) << 8) | (uint32_t)cur;   // (y, x, score): sort key
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-cell dual threshold and ordering (src/ORBextractor.cc:799-831): if any survivor of the cell reaches iniThFAST only those are kept
// (cv::FAST with iniThFAST returned something), otherwise all of them (the minThFAST retry); cv::FAST returns key points in row-major
// order.  One wave per cell sorts the (y, x, score) keys in LDS and rewrites the slot as (x, y, score) entries.
constexpr int CS_MAX = 1024;    // a cell is < 60 px wide and high (ceil(width / (width / 30))), so at most 30 x 30 strict local maxima
__global__ __launch_bounds__(64) void k_cells_sort(const OrbGeom* __restrict__ gp, uint32_t* __restrict__ cells, int* __restrict__ cellCount)
{
    OLF_SET_GUEST_PRIO();
    __shared__ uint32_t s_k[CS_MAX];
    const OrbGeom& g = *gp;
    const int img = blockIdx.y, cell = blockIdx.x, lane = threadIdx.x;
    int* cnt = cellCount + (size_t)img * g.totalCells + cell;
    const int n = min(*cnt, g.cellCap);
    if (n == 0) return;
    uint32_t* slot = cells + ((size_t)img * g.totalCells + cell) * g.cellCap;
    if (n <= 64) {
        // the common case: one key per lane, rank = number of smaller keys (keys are distinct: one per pixel), no LDS, no barriers
        uint32_t k = lane < n ? slot[lane] : 0xffffffffu;
        const bool anyIni = wave_vote(lane < n && (int)(k & 0xffu) >= g.iniTh) != 0;
        if (anyIni && (int)(k & 0xffu) < g.iniTh) k = 0xffffffffu;
        const int kept = __popcll(wave_vote(k != 0xffffffffu));
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (uint32_t)__builtin_amdgcn_readlane((int)k, j) < k;
        if (k != 0xffffffffu) slot[rank] = (((k >> 8) & 0xfffu) << 20) | ((k >> 20) << 8) | (k & 0xffu);
        if (lane == 0) *cnt = kept;
        return;
    }
    int sortN = 128;
    while (sortN < n) sortN <<= 1;
    bool anyIni = false;
    for (int i = lane; i < sortN; i += 64) {
        const uint32_t k = i < n ? slot[i] : 0xffffffffu;
        s_k[i] = k;
        anyIni |= i < n && (int)(k & 0xffu) >= g.iniTh;
    }
    anyIni = wave_vote(anyIni) != 0;
    __builtin_amdgcn_wave_barrier();
    int kept = n;
    if (anyIni) {
        int drop = 0;
        for (int i = lane; i < n; i += 64)
            if ((int)(s_k[i] & 0xffu) < g.iniTh) { s_k[i] = 0xffffffffu; ++drop; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) drop += __shfl_xor(drop, o);
        kept = n - drop;
        __builtin_amdgcn_wave_barrier();
    }
    for (int k = 2; k <= sortN; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < sortN; t += 64) {
                const int p = t ^ j;
                if (p > t) {
                    const uint32_t a = s_k[t], b = s_k[p];
                    if (((t & k) == 0) ? a > b : a < b) { s_k[t] = b; s_k[p] = a; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    for (int i = lane; i < kept; i += 64) {
        const uint32_t k = s_k[i];
        slot[i] = (((k >> 8) & 0xfffu) << 20) | ((k >> 20) << 8) | (k & 0xffu);
    }
    if (lane == 0) *cnt = kept;
}

// ---------------------------------------------------------------------------------------------
int launch_orb_pyramid(const OrbGeom& g, const OrbDeviceBufs& b, const uint8_t* d_in, int n_images, hipStream_t s)
{
    {
        const int quads = ((g.W + 15) >> 4) * g.H;
        hipLaunchKernelGGL(k_ingest, dim3((quads + 255) / 256, n_images), dim3(256), 0, s, d_in, b.pyr, g.W, g.H, g.in_pitch,
                           g.lv[0].pitch, g.pyrBytes);
    }
    for (int l = 1; l < g.nlevels; ++l) {
        const LevelGeom &P = g.lv[l - 1], &L = g.lv[l];
        if (L.resizeTiled) {
            int rc = launch_resize_tiled(b.pyr + P.offset, (size_t)g.pyrBytes, P.pitch, P.w, P.h, b.pyr + L.offset, (size_t)g.pyrBytes, L.pitch, L.w, L.h,
                                         b.rx + L.resizeTabX, b.ry + L.resizeTabY, n_images, s, (L.resizeTiled & 2) != 0);
            if (rc != OLF_OK) return rc;
        } else {
            const int quads = ((L.w + 3) >> 2) * L.h;
            hipLaunchKernelGGL(k_resize, dim3((quads + 255) / 256, n_images), dim3(256), 0, s, b.pyr, g.pyrBytes, P, L, b.rx, b.ry);
        }
    }
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_orb_fast(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, hipStream_t s)
{
    OLF_HIP_CHECK(hipMemsetAsync(b.cellCount, 0, (size_t)n_images * g.totalCells * sizeof(int), s));
    for (int l = 0; l < g.nlevels; ++l) {
        const LevelGeom& L = g.lv[l];
        const int fw = L.maxBorderX - 3 - (kMinBorder - 2), fh = L.maxBorderY - 3 - (kMinBorder + 3);   // emitted columns start at 14, rows at 19
        if (fw <= 0 || fh <= 0) continue;
        static const int nt = [] { const char* e = getenv("OLF_FAST_NT"); const int v = e ? atoi(e) : 256; return v == 128 ? 128 : 256; }();
        if (nt == 128)
            hipLaunchKernelGGL(k_fast_score<128>, dim3((fw + FT_EW - 1) / FT_EW, (fh + 14 - 1) / 14, n_images), dim3(128), 0, s, b.pyr,
                               g.pyrBytes, L, g.minTh, b.cells, b.cellCount, g.totalCells, g.cellCap);
        else
            hipLaunchKernelGGL(k_fast_score<256>, dim3((fw + FT_EW - 1) / FT_EW, (fh + 30 - 1) / 30, n_images), dim3(256), 0, s, b.pyr,
                               g.pyrBytes, L, g.minTh, b.cells, b.cellCount, g.totalCells, g.cellCap);
    }
    hipLaunchKernelGGL(k_cells_sort, dim3(g.totalCells, n_images), dim3(64), 0, s, b.geom, b.cells, b.cellCount);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_orb_blur(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, hipStream_t s)
{
    for (int l = 0; l < g.nlevels; ++l) {
        const LevelGeom& L = g.lv[l];
        int rc = launch_sep7(b.pyr + L.offset, (size_t)g.pyrBytes, L.pitch, b.blur + L.offset, (size_t)g.pyrBytes, L.pitch, L.w, L.h, g.blurTaps,
                             n_images, s);
        if (rc != OLF_OK) return rc;
    }
    return OLF_OK;
}

}  // namespace olf
