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
    const int img = blockIdx.y;
    const int quads = (W + 3) >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= quads * H) return;
    const int y = idx / quads, x = (idx - y * quads) * 4;
    const uint8_t* src = in + (size_t)img * H * in_pitch + (size_t)y * in_pitch + x;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (x + k < W) v |= (uint32_t)src[k] << (8 * k);
    *reinterpret_cast<uint32_t*>(pyr + (size_t)img * pyrBytes + (size_t)y * pitch0 + x) = v;
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
__device__ __forceinline__ int arc9_maxmin(const int* d)
{
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m2[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m8[i] = min(m4[i], m4[(i + 4) & 15]);
    int best = -1000;
#pragma unroll
    for (int i = 0; i < 16; ++i) best = max(best, min(m8[i], d[(i + 8) & 15]));
    return best;
}

__device__ __forceinline__ bool run9(uint32_t m)   // >= 9 contiguous set bits in a circular 16-bit mask
{
    uint32_t x = m | (m << 16);
    uint32_t r = x & (x >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= x >> 8;
    return (r & 0xffffu) != 0;
}

constexpr int FT_W = 64, FT_H = 16, FT_PITCH = 72;   // LDS tile (64+6) x (16+6), pitch 72

__global__ __launch_bounds__(256) void k_fast_score(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ score,
                                                    int pyrBytes, LevelGeom L, int minTh)
{
    __shared__ uint8_t tile[(FT_H + 6) * FT_PITCH];
    const int img = blockIdx.z;
    // scores are needed on x in [minBorder+3, maxBorderX-3), y likewise
    const int x0 = kMinBorder + 3 + blockIdx.x * FT_W, y0 = kMinBorder + 3 + blockIdx.y * FT_H;
    const int xEnd = L.maxBorderX - 3, yEnd = L.maxBorderY - 3;
    const uint8_t* src = pyr + (size_t)img * pyrBytes + L.offset;
    for (int i = threadIdx.x; i < (FT_H + 6) * (FT_W + 6); i += 256) {
        const int ty = i / (FT_W + 6), tx = i - ty * (FT_W + 6);
        const int gx = min(x0 - 3 + tx, L.w - 1), gy = min(y0 - 3 + ty, L.h - 1);
        tile[ty * FT_PITCH + tx] = src[(size_t)gy * L.pitch + gx];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
    uint8_t* dst = score + (size_t)img * pyrBytes + L.offset;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ly = ly0 + 4 * r;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= xEnd || gy >= yEnd) continue;
        const uint8_t* c = &tile[(ly + 3) * FT_PITCH + lx + 3];
        const int v = c[0];
        int ring[16];
        ring[0] = c[3 * FT_PITCH];       ring[1] = c[3 * FT_PITCH + 1];   ring[2] = c[2 * FT_PITCH + 2];
        ring[3] = c[FT_PITCH + 3];       ring[4] = c[3];                  ring[5] = c[-FT_PITCH + 3];
        ring[6] = c[-2 * FT_PITCH + 2];  ring[7] = c[-3 * FT_PITCH + 1];  ring[8] = c[-3 * FT_PITCH];
        ring[9] = c[-3 * FT_PITCH - 1];  ring[10] = c[-2 * FT_PITCH - 2]; ring[11] = c[-FT_PITCH - 3];
        ring[12] = c[-3];                ring[13] = c[FT_PITCH - 3];      ring[14] = c[2 * FT_PITCH - 2];
        ring[15] = c[3 * FT_PITCH - 1];
        uint32_t dark = 0, bright = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            dark |= (uint32_t)(ring[k] < v - minTh) << k;
            bright |= (uint32_t)(ring[k] > v + minTh) << k;
        }
        int s = 0;
        if (run9(dark) || run9(bright)) {
            int d[16], nd[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { d[k] = v - ring[k]; nd[k] = -d[k]; }
            s = max(arc9_maxmin(d), arc9_maxmin(nd)) - 1;
            if (s < minTh) s = 0;
        }
        dst[(size_t)gy * L.pitch + gx] = (uint8_t)s;
    }
}

// ---------------------------------------------------------------------------------------------
// Per-cell detection (src/ORBextractor.cc:791-831): cv::FAST(cell sub-image, iniThFAST, nms) and,
// if that returns nothing, cv::FAST(..., minThFAST, nms).  With the score map this is: take the
// strict 3x3 local maxima of the score restricted to the cell's interior (the sub-image minus its
// 3-pixel frame; everything outside counts as 0); emit those with score >= iniTh if any exist,
// otherwise all of them.  One wave per cell; output row-major inside the cell slot.
__global__ __launch_bounds__(64) void k_cells(const uint8_t* __restrict__ score, const OrbGeom* __restrict__ gp,
                                              uint32_t* __restrict__ cells, int* __restrict__ cellCount)
{
    __shared__ uint8_t t[64 * 64];                // (ih+2) x (iw+2), pitch 64
    __shared__ unsigned long long keepM[64], iniM[64];
    __shared__ int rowBase[64];
    const OrbGeom& g = *gp;
    const int img = blockIdx.y, cell = blockIdx.x, lane = threadIdx.x;
    int l = 0;
    while (l + 1 < g.nlevels && cell >= g.lv[l + 1].cellBase) ++l;
    const LevelGeom& L = g.lv[l];
    const int ci = (cell - L.cellBase) / L.nCols, cj = (cell - L.cellBase) - ci * L.nCols;
    const int iniX = kMinBorder + cj * L.wCell, iniY = kMinBorder + ci * L.hCell;
    int* cnt = cellCount + (size_t)img * g.totalCells + cell;
    if (iniY >= L.maxBorderY - 3 || iniX >= L.maxBorderX - 6) { if (lane == 0) *cnt = 0; return; }
    const int maxX = min(iniX + L.wCell + 6, L.maxBorderX), maxY = min(iniY + L.hCell + 6, L.maxBorderY);
    const int ix0 = iniX + 3, iy0 = iniY + 3, iw = maxX - 3 - ix0, ih = maxY - 3 - iy0;
    if (iw <= 0 || ih <= 0) { if (lane == 0) *cnt = 0; return; }
    const uint8_t* src = score + (size_t)img * g.pyrBytes + L.offset;
    for (int i = lane; i < (ih + 2) * 64; i += 64) {
        const int ty = i >> 6, tx = i & 63;
        uint8_t v = 0;
        if (tx >= 1 && tx <= iw && ty >= 1 && ty <= ih) v = src[(size_t)(iy0 + ty - 1) * L.pitch + ix0 + tx - 1];
        t[i] = v;
    }
    __syncthreads();
    unsigned long long anyIni = 0;
    for (int y = 0; y < ih; ++y) {
        bool keep = false, ini = false;
        if (lane < iw) {
            const uint8_t* c = &t[(y + 1) * 64 + lane + 1];
            const int s = c[0];
            keep = s > 0 && s > c[-1] && s > c[1] && s > c[-65] && s > c[-64] && s > c[-63] && s > c[63] && s > c[64] && s > c[65];
            ini = keep && s >= g.iniTh;
        }
        const unsigned long long km = __ballot(keep), im = __ballot(ini);
        if (lane == 0) { keepM[y] = km; iniM[y] = im; }
        anyIni |= im;
    }
    __syncthreads();
    const unsigned long long* chosen = anyIni ? iniM : keepM;
    int c = (lane < ih) ? __popcll(chosen[lane]) : 0;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
    }
    rowBase[lane] = inc - c;
    const int total = __shfl(inc, 63);
    __syncthreads();
    uint32_t* slot = cells + ((size_t)img * g.totalCells + cell) * g.cellCap;
    for (int y = 0; y < ih; ++y) {
        const unsigned long long m = chosen[y];
        if (lane < iw && ((m >> lane) & 1ull)) {
            const int pos = rowBase[y] + __popcll(m & ((1ull << lane) - 1ull));
            const int s = t[(y + 1) * 64 + lane + 1];
            slot[pos] = ((uint32_t)(ix0 + lane - kMinBorder) << 20) | ((uint32_t)(iy0 + y - kMinBorder) << 8) | (uint32_t)s;
        }
    }
    if (lane == 0) *cnt = total;
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur(7x7, sigma=2, BORDER_REFLECT_101) in 8-bit fixed point (App. A.3): row pass exact
// (fits u16: 257*255), column pass (sum + 2^15) >> 16 saturated.
constexpr int BT_W = 64, BT_H = 16;

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p;
}

__global__ __launch_bounds__(256) void k_blur7(const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                               const OrbGeom* __restrict__ gp, int level)
{
    __shared__ uint8_t in[(BT_H + 6) * 72];
    __shared__ uint16_t hrow[(BT_H + 6) * 64];
    const OrbGeom& g = *gp;
    const LevelGeom& L = g.lv[level];
    const int img = blockIdx.z;
    const int x0 = blockIdx.x * BT_W, y0 = blockIdx.y * BT_H;
    const uint8_t* src = pyr + (size_t)img * g.pyrBytes + L.offset;
    for (int i = threadIdx.x; i < (BT_H + 6) * (BT_W + 6); i += 256) {
        const int ty = i / (BT_W + 6), tx = i - ty * (BT_W + 6);
        const int gx = reflect101(min(x0 - 3 + tx, L.w + 2), L.w), gy = reflect101(min(y0 - 3 + ty, L.h + 2), L.h);
        in[ty * 72 + tx] = src[(size_t)gy * L.pitch + gx];
    }
    __syncthreads();
    int taps[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) taps[k] = g.blurTaps[k];
    for (int i = threadIdx.x; i < (BT_H + 6) * BT_W; i += 256) {
        const int ty = i >> 6, tx = i & 63;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += taps[k] * in[ty * 72 + tx + k];
        hrow[i] = (uint16_t)acc;
    }
    __syncthreads();
    uint8_t* dst = blur + (size_t)img * g.pyrBytes + L.offset;
    const int lx = threadIdx.x & 63, ly0 = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ly = ly0 + 4 * r;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= L.w || gy >= L.h) continue;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += taps[k] * hrow[(ly + k) * 64 + lx];
        acc = (acc + 32768) >> 16;
        dst[(size_t)gy * L.pitch + gx] = (uint8_t)min(acc, 255);
    }
}

// ---------------------------------------------------------------------------------------------
int launch_orb_pyramid(const OrbGeom& g, const OrbDeviceBufs& b, const uint8_t* d_in, int n_images, hipStream_t s)
{
    {
        const int quads = ((g.W + 3) >> 2) * g.H;
        hipLaunchKernelGGL(k_ingest, dim3((quads + 255) / 256, n_images), dim3(256), 0, s, d_in, b.pyr, g.W, g.H, g.in_pitch,
                           g.lv[0].pitch, g.pyrBytes);
    }
    for (int l = 1; l < g.nlevels; ++l) {
        const LevelGeom &P = g.lv[l - 1], &L = g.lv[l];
        if (L.resizeTiled) {
            int rc = launch_resize_tiled(b.pyr + P.offset, (size_t)g.pyrBytes, P.pitch, P.w, P.h, b.pyr + L.offset, (size_t)g.pyrBytes, L.pitch, L.w, L.h,
                                         b.rx + L.resizeTabX, b.ry + L.resizeTabY, n_images, s);
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
    for (int l = 0; l < g.nlevels; ++l) {
        const LevelGeom& L = g.lv[l];
        const int fw = L.maxBorderX - 3 - (kMinBorder + 3), fh = L.maxBorderY - 3 - (kMinBorder + 3);
        if (fw <= 0 || fh <= 0) continue;
        hipLaunchKernelGGL(k_fast_score, dim3((fw + FT_W - 1) / FT_W, (fh + FT_H - 1) / FT_H, n_images), dim3(256), 0, s, b.pyr,
                           b.score, g.pyrBytes, L, g.minTh);
    }
    hipLaunchKernelGGL(k_cells, dim3(g.totalCells, n_images), dim3(64), 0, s, b.score, b.geom, b.cells, b.cellCount);
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
