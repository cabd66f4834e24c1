// filters.hip -- the 8-bit fixed-point separable Gaussian shared by the path, gfx950:
//   GaussianBlur(7x7, sigma 2)   per pyramid level, reference src/ORBextractor.cc:1087-1088
//   GaussianBlur(7x7, sigma 0.6) inside cv::LineSegmentDetector (App. A.7 step 1)
//   GaussianBlur(5x5, sigma 1)   BinaryDescriptor::computeGaussianPyramid, binary_descriptor_custom.cpp:358
// Row pass exact in u16 (taps sum <= 257, 257*255 = 65535), column pass (sum + 2^15) >> 16 saturated, BORDER_REFLECT_101.
// HBM-bound: each block reads a (128+8) x (32+6) byte halo tile with 4-byte loads (v_alignbit for rows that are not
// 4-byte aligned), filters 4 pixels per thread out of LDS words and writes 4-byte words: one read + one write of the image.
#include "olf_internal.hpp"
#include <algorithm>

namespace olf {

constexpr int SF_TW = 128, SF_TH = 32;                 // output tile
constexpr int SF_INW = (SF_TW + 8) / 4;                // input words per tile row (x0-4 .. x0+TW+3)
constexpr int SF_INH = SF_TH + 6;

__device__ __forceinline__ int sf_reflect(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p;
}

__global__ __launch_bounds__(256) void k_sep7(const uint8_t* __restrict__ src, size_t srcImgStride, int srcPitch, uint8_t* __restrict__ dst,
                                              size_t dstImgStride, int dstPitch, int W, int H, Taps7 taps)
{
    __shared__ uint32_t s_in[SF_INH * SF_INW];                 // bytes x0-4 .. x0+131 of rows y0-3 .. y0+34
    // row-pass results, column major: column x holds its SF_INH u16 values as row pairs (rows 2j, 2j+1 in dword j); the odd column
    // stride keeps the column pass's LDS reads spread over the banks
    constexpr int HS = (SF_INH + 1) / 2 + ((((SF_INH + 1) / 2) & 1) ? 0 : 1);
    __shared__ uint32_t s_h[SF_TW * HS];
    const int img = blockIdx.z, x0 = blockIdx.x * SF_TW, y0 = blockIdx.y * SF_TH;
    const uint8_t* s = src + (size_t)img * srcImgStride;
    // interior tile of a 4-byte aligned image: every staged word is one aligned load, no reflection (the common case by far)
    const bool plain = x0 >= 4 && x0 + SF_TW + 4 <= W && y0 >= 3 && y0 + SF_TH + 3 <= H && ((srcPitch | (int)srcImgStride) & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(src) & 3) == 0;
    const uint8_t* tile = s + (ptrdiff_t)(y0 - 3) * srcPitch + (x0 - 4);
    for (int i = threadIdx.x; i < SF_INH * SF_INW; i += 256) {
        const int r = i / SF_INW, j = i - r * SF_INW;
        uint32_t v;
        if (plain) v = *reinterpret_cast<const uint32_t*>(tile + (size_t)r * srcPitch + 4 * j);
        else {
            const int gy = sf_reflect(min(y0 - 3 + r, H + 2), H);
            const int xw = x0 - 4 + 4 * j;
            const uint8_t* row = s + (size_t)gy * srcPitch;
            const uintptr_t addr = reinterpret_cast<uintptr_t>(row + xw);
            const int sh = (int)(addr & 3) * 8;
            if (xw >= 0 && (sh ? xw + 7 < W : xw + 3 < W)) {          // the second aligned word must stay inside the row
                const uint32_t* p4 = reinterpret_cast<const uint32_t*>(row + xw - (int)(addr & 3));      // (pointer arithmetic, not an integer cast: stays a global load)
                v = sh ? __funnelshift_r(p4[0], p4[1], sh) : p4[0];
            } else {
                v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) v |= (uint32_t)row[sf_reflect(min(xw + b, W + 2), W)] << (8 * b);
            }
        }
        s_in[i] = v;
    }
    __syncthreads();
    // ---- row pass: 4 outputs per task from bytes [4q+1, 4q+10] of the tile row (tile byte 4 == image x0).  The taps are 8-bit
    // fractions (OpenCV's ufixedpoint16 row filter), so a 7-tap row sum is two v_dot4_u32_u8 on byte windows cut out with v_alignbyte.
    const uint32_t tlo = (uint32_t)taps.t[0] | ((uint32_t)taps.t[1] << 8) | ((uint32_t)taps.t[2] << 16) | ((uint32_t)taps.t[3] << 24);
    const uint32_t thi = (uint32_t)taps.t[4] | ((uint32_t)taps.t[5] << 8) | ((uint32_t)taps.t[6] << 16);
    uint16_t* s_h16 = reinterpret_cast<uint16_t*>(s_h);
    for (int i = threadIdx.x; i < SF_INH * (SF_TW / 4); i += 256) {
        const int r = i / (SF_TW / 4), q = i - r * (SF_TW / 4);
        const uint32_t w0 = s_in[r * SF_INW + q], w1 = s_in[r * SF_INW + q + 1], w2 = s_in[r * SF_INW + q + 2];
        // output x = x0+4q+p uses x-3 .. x+3 = tile bytes 4q+p+1 .. 4q+p+7
        const uint32_t o0 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1u), thi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1u), tlo, 0u, false), false);
        const uint32_t o1 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2u), thi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2u), tlo, 0u, false), false);
        const uint32_t o2 = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3u), thi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3u), tlo, 0u, false), false);
        const uint32_t o3 = __builtin_amdgcn_udot4(w2, thi, __builtin_amdgcn_udot4(w1, tlo, 0u, false), false);
        uint16_t* col = s_h16 + (size_t)(4 * q) * (2 * HS) + r;
        col[0] = (uint16_t)o0; col[2 * HS] = (uint16_t)o1; col[4 * HS] = (uint16_t)o2; col[6 * HS] = (uint16_t)o3;
    }
    __syncthreads();
    // ---- column pass: each thread 4 pixels wide x 4 rows tall.  Rows ry0 .. ry0+9 of a column are 5 row-pair dwords D0..D4; an even
    // output row uses the pairs as stored, an odd one the pairs shifted by one row (v_alignbyte); 7 taps = 4 v_dot2_u32_u16, the
    // rounding constant rides in the first accumulator.
    const int q = threadIdx.x & 31, ry0 = (threadIdx.x >> 5) * 4;
    const int gx = x0 + 4 * q;
    if (gx >= W) return;
    const uint32_t t01 = (uint32_t)taps.t[0] | ((uint32_t)taps.t[1] << 16), t23 = (uint32_t)taps.t[2] | ((uint32_t)taps.t[3] << 16);
    const uint32_t t45 = (uint32_t)taps.t[4] | ((uint32_t)taps.t[5] << 16), t6lo = (uint32_t)taps.t[6], t6hi = (uint32_t)taps.t[6] << 16;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto dot2 = [](uint32_t a, uint32_t b, uint32_t c) -> uint32_t {
        us2 va, vb;
        __builtin_memcpy(&va, &a, 4); __builtin_memcpy(&vb, &b, 4);
        return __builtin_amdgcn_udot2(va, vb, c, false);
    };
    uint32_t res[4][4];        // [row][column]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t* hp = s_h + (size_t)(4 * q + c) * HS + (ry0 >> 1);
        const uint32_t D0 = hp[0], D1 = hp[1], D2 = hp[2], D3 = hp[3], D4 = hp[4];
        const uint32_t E0 = __builtin_amdgcn_alignbyte(D1, D0, 2u), E1 = __builtin_amdgcn_alignbyte(D2, D1, 2u);
        const uint32_t E2 = __builtin_amdgcn_alignbyte(D3, D2, 2u), E3 = __builtin_amdgcn_alignbyte(D4, D3, 2u);
        res[0][c] = dot2(D3, t6lo, dot2(D2, t45, dot2(D1, t23, dot2(D0, t01, 32768u))));      // rows 0..6
        res[1][c] = dot2(D3, t6hi, dot2(E2, t45, dot2(E1, t23, dot2(E0, t01, 32768u))));      // rows 1..7
        res[2][c] = dot2(D4, t6lo, dot2(D3, t45, dot2(D2, t23, dot2(D1, t01, 32768u))));      // rows 2..8
        res[3][c] = dot2(D4, t6hi, dot2(E3, t45, dot2(E2, t23, dot2(E1, t01, 32768u))));      // rows 3..9
    }
    uint8_t* d = dst + (size_t)img * dstImgStride;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int gy = y0 + ry0 + rr;
        if (gy >= H) break;
        const uint32_t a0 = min(res[rr][0] >> 16, 255u), a1 = min(res[rr][1] >> 16, 255u), a2 = min(res[rr][2] >> 16, 255u), a3 = min(res[rr][3] >> 16, 255u);
        uint8_t* o = d + (size_t)gy * dstPitch + gx;
        if (gx + 3 < W && ((reinterpret_cast<uintptr_t>(o) & 3) == 0))
            *reinterpret_cast<uint32_t*>(o) = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
        else {
            o[0] = (uint8_t)a0;
            if (gx + 1 < W) o[1] = (uint8_t)a1;
            if (gx + 2 < W) o[2] = (uint8_t)a2;
            if (gx + 3 < W) o[3] = (uint8_t)a3;
        }
    }
}

// ---- the same filter without LDS (round 4): one thread = a strip of 4 columns x SS_ROWS rows, everything in registers.  Per input row three dwords
// (bytes x0-4 .. x0+7; rows of neighbouring strips overlap in L1 / L2), the 7-tap row sums of the four columns by v_dot4_u32_u8 on byte windows, row pairs
// packed into dwords, the column sums by v_dot2_u32_u16 -- the arithmetic of k_sep7, 13 instead of 32 vector instructions per pixel: no staging pass, no
// u16 round trip through LDS, no barriers (which also makes it a better guest beside the growth agents: a block's time is its own loads, nothing else).
// BORDER_REFLECT_101 in y is an index computation per row; the strips whose window leaves the image in x (strip 0, strips beyond nsx) are a second, small
// launch of the same kernel with a byte-wise row window (k_sep7_strip<true>).
constexpr int SS_ROWS = 16;
#ifndef OLF_SS_BROWS
#define OLF_SS_BROWS 8
#endif
constexpr int SS_BROWS = OLF_SS_BROWS;       // rows per thread of the border strips (k_sep7_strip)

__device__ __forceinline__ int ss_reflect(int p, int n) { p = p < 0 ? -p : p; return p >= n ? 2 * (n - 1) - p : p; }

// BORDER: the strips whose window leaves the image -- strip 0 and the strips right of nsx -- with the row window assembled byte by byte under
// BORDER_REFLECT_101 and the quad stored byte by byte; a separate small launch, so that no wave of the interior pays for it.
template <bool BORDER, int ROWS>
__device__ __forceinline__ void sep7_strip_body(const int t, const uint8_t* __restrict__ src, size_t srcImgStride, int srcPitch, uint8_t* __restrict__ dst,
                                                size_t dstImgStride, int dstPitch, int W, int H, const Taps7& taps, int nsx, int nsy)
{
    const int nb = BORDER ? 1 + ((W + 3) / 4 - (nsx + 1)) : nsx;      // strips per strip row of this part
    if (t >= nb * nsy) return;
    const int sy = t / nb, si = t - sy * nb;
    const int sx = BORDER ? (si == 0 ? 0 : nsx + si) : si + 1;
    const int x0 = 4 * sx, y0 = sy * ROWS;
    const uint8_t* s = src + (size_t)blockIdx.y * srcImgStride + (x0 - 4);
    uint8_t* d = dst + (size_t)blockIdx.y * dstImgStride + x0;
    const uint32_t tlo = (uint32_t)taps.t[0] | ((uint32_t)taps.t[1] << 8) | ((uint32_t)taps.t[2] << 16) | ((uint32_t)taps.t[3] << 24);
    const uint32_t thi = (uint32_t)taps.t[4] | ((uint32_t)taps.t[5] << 8) | ((uint32_t)taps.t[6] << 16);
    const uint32_t t01 = (uint32_t)taps.t[0] | ((uint32_t)taps.t[1] << 16), t23 = (uint32_t)taps.t[2] | ((uint32_t)taps.t[3] << 16);
    const uint32_t t45 = (uint32_t)taps.t[4] | ((uint32_t)taps.t[5] << 16), t6lo = (uint32_t)taps.t[6], t6hi = (uint32_t)taps.t[6] << 16;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto dot2 = [](uint32_t a, uint32_t b, uint32_t c) -> uint32_t {
        us2 va, vb;
        __builtin_memcpy(&va, &a, 4); __builtin_memcpy(&vb, &b, 4);
        return __builtin_amdgcn_udot2(va, vb, c, false);
    };
    // one input row of the strip: 12 bytes x0-4 .. x0+7 of image row y (reflected into the image; rows beyond H + 2 only feed output rows that do not exist)
    struct Raw { uint32_t w[3]; };
    auto load_row = [&](int y) -> Raw {
        Raw r;
        const uint8_t* row = s + (size_t)ss_reflect(min(y, H + 2), H) * srcPitch;
        if (!BORDER) __builtin_memcpy(r.w, row, 12);
        else {
            // (row points at column x0 - 4; columns beyond W + 2 only feed outputs that do not exist)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int x = x0 - 4 + 4 * k + b;
                    v |= (uint32_t)row[ss_reflect(min(x, W + 2), W) - (x0 - 4)] << (8 * b);
                }
                r.w[k] = v;
            }
        }
        return r;
    };
    // the 7-tap row sums (u16 range) of the strip's four columns: output x0 + p uses bytes p + 1 .. p + 7 of the 12
    auto hrow = [&](const Raw& r, uint32_t* o) {
        const uint32_t* w = r.w;
        o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w[2], w[1], 1u), thi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w[1], w[0], 1u), tlo, 0u, false), false);
        o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w[2], w[1], 2u), thi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w[1], w[0], 2u), tlo, 0u, false), false);
        o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w[2], w[1], 3u), thi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w[1], w[0], 3u), tlo, 0u, false), false);
        o[3] = __builtin_amdgcn_udot4(w[2], thi, __builtin_amdgcn_udot4(w[1], tlo, 0u, false), false);
    };
    // pair j = the row sums of image rows y0 - 3 + 2j (low half) and y0 - 2 + 2j (high half), per column
    auto pair = [&](const Raw& ra, const Raw& rb, uint32_t* Pj) {
        uint32_t a[4], b[4];
        hrow(ra, a); hrow(rb, b);
#pragma unroll
        for (int c = 0; c < 4; ++c) Pj[c] = a[c] | (b[c] << 16);
    };
    uint32_t P[ROWS / 2 + 3][4];
    Raw nx[4];      // the four input rows of the NEXT group of output rows: requested one group ahead, so that their latency (and the acknowledgement of the
                    // stores in between: one counter for both on this part) is covered by a group's arithmetic
    {
        Raw r0 = load_row(y0 - 3), r1 = load_row(y0 - 2), r2 = load_row(y0 - 1), r3 = load_row(y0), r4 = load_row(y0 + 1), r5 = load_row(y0 + 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) nx[k] = load_row(y0 + 3 + k);
        pair(r0, r1, P[0]); pair(r2, r3, P[1]); pair(r4, r5, P[2]);
    }
#pragma unroll
    for (int g = 0; g < ROWS / 4; ++g) {
        // output rows y0 + 4g .. y0 + 4g + 3 need image rows y0 + 4g - 3 .. y0 + 4g + 6 = pairs 2g .. 2g + 4
        const Raw c0 = nx[0], c1 = nx[1], c2 = nx[2], c3 = nx[3];
        if (g + 1 < ROWS / 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) nx[k] = load_row(y0 + 4 * (g + 1) + 3 + k);
            __builtin_amdgcn_sched_barrier(0);      // (the requests stay in front of the arithmetic below)
        }
        pair(c0, c1, P[2 * g + 3]); pair(c2, c3, P[2 * g + 4]);
        uint32_t res[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t D0 = P[2 * g][c], D1 = P[2 * g + 1][c], D2 = P[2 * g + 2][c], D3 = P[2 * g + 3][c], D4 = P[2 * g + 4][c];
            const uint32_t E0 = __builtin_amdgcn_alignbyte(D1, D0, 2u), E1 = __builtin_amdgcn_alignbyte(D2, D1, 2u);
            const uint32_t E2 = __builtin_amdgcn_alignbyte(D3, D2, 2u), E3 = __builtin_amdgcn_alignbyte(D4, D3, 2u);
            res[0][c] = dot2(D3, t6lo, dot2(D2, t45, dot2(D1, t23, dot2(D0, t01, 32768u))));
            res[1][c] = dot2(D3, t6hi, dot2(E2, t45, dot2(E1, t23, dot2(E0, t01, 32768u))));
            res[2][c] = dot2(D4, t6lo, dot2(D3, t45, dot2(D2, t23, dot2(D1, t01, 32768u))));
            res[3][c] = dot2(D4, t6hi, dot2(E3, t45, dot2(E2, t23, dot2(E1, t01, 32768u))));
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int gy = y0 + 4 * g + rr;
            if (gy < H) {
                const uint32_t a0 = min(res[rr][0] >> 16, 255u), a1 = min(res[rr][1] >> 16, 255u), a2 = min(res[rr][2] >> 16, 255u), a3 = min(res[rr][3] >> 16, 255u);
                const uint32_t v = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
                if (!BORDER) __builtin_memcpy(d + (size_t)gy * dstPitch, &v, 4);
                else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) if (x0 + b < W) d[(size_t)gy * dstPitch + b] = (uint8_t)(v >> (8 * b));
                }
            }
        }
    }
}

// The interior strips (16 rows per thread, 70 VGPRs) and the strips whose window leaves the image in x (a byte-wise row window; SS_BROWS = 8 rows per thread: twice
// the threads and half the dependent chain of the 16-row form, which took 0.21 ms alone and up to 22 ms in the seed sort's shadow, where 72 threads per
// image get no issue slots and everything behind them on the ORB stream waits; same-box step times 227.0 / 227.0 / 226.7 ms with 8 rows, 227.3 / 227.3 / 228.3
// with 4, 227.5 / 227.7 / 227.6 with 16: profiles/r4ag_border_strip_rows_ab.txt).  Two launches: as two halves of one kernel the interior strips
// inherit the border part's 145 VGPRs (two waves per SIMD).
template <bool BORDER>
__global__ __launch_bounds__(256) void k_sep7_strip(const uint8_t* __restrict__ src, size_t srcImgStride, int srcPitch, uint8_t* __restrict__ dst,
                                                    size_t dstImgStride, int dstPitch, int W, int H, Taps7 taps, int nsx, int nsy)
{
    sep7_strip_body<BORDER, BORDER ? SS_BROWS : SS_ROWS>(blockIdx.x * 256 + threadIdx.x, src, srcImgStride, srcPitch, dst, dstImgStride, dstPitch, W, H, taps, nsx, nsy);
}

int launch_sep7(const uint8_t* src, size_t srcImgStride, int srcPitch, uint8_t* dst, size_t dstImgStride, int dstPitch, int W, int H,
                const int* taps7, int n_images, hipStream_t s)
{
    Taps7 t;
    for (int i = 0; i < 7; ++i) t.t[i] = taps7[i];
    if (t.t[3] == 256) {
        // the identity kernel (cv::LineSegmentDetector at scale 1 skips its blur; a 256 does not fit the 8-bit tap fields of k_sep7): pitched copy
        for (int i = 0; i < n_images; ++i)
            OLF_HIP_CHECK(hipMemcpy2DAsync(dst + (size_t)i * dstImgStride, dstPitch, src + (size_t)i * srcImgStride, srcPitch, W, H, hipMemcpyDeviceToDevice, s));
        return OLF_OK;
    }
    for (int i = 0; i < 7; ++i)
        if (t.t[i] < 0 || t.t[i] > 255) { set_error("launch_sep7: taps must be 8-bit fractions"); return OLF_ERR_INVALID; }
    // OLF_SEP7=0: the LDS-tiled kernel of rounds 1-3 (A/B measurements); images too small for a strip (or a reflection that would leave them) take it too
    static const bool strips = [] { const char* e = getenv("OLF_SEP7"); return !e || atoi(e) != 0; }();
    if (strips && W >= 16 && H >= 8 && (dstPitch & 3) == 0 && (dstImgStride & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
        const int nsx = (W - 7) / 4, nsy = (H + SS_ROWS - 1) / SS_ROWS, nb = 1 + ((W + 3) / 4 - (nsx + 1));
        const int nsyB = (H + SS_BROWS - 1) / SS_BROWS;
        hipLaunchKernelGGL(k_sep7_strip<false>, dim3((nsx * nsy + 255) / 256, n_images), dim3(256), 0, s, src, srcImgStride, srcPitch, dst, dstImgStride, dstPitch, W, H, t, nsx, nsy);
        hipLaunchKernelGGL(k_sep7_strip<true>, dim3((nb * nsyB + 255) / 256, n_images), dim3(256), 0, s, src, srcImgStride, srcPitch, dst, dstImgStride, dstPitch, W, H, t, nsx, nsyB);
        OLF_HIP_CHECK(hipGetLastError());
        return OLF_OK;
    }
    hipLaunchKernelGGL(k_sep7, dim3((W + SF_TW - 1) / SF_TW, (H + SF_TH - 1) / SF_TH, n_images), dim3(256), 0, s, src, srcImgStride, srcPitch,
                       dst, dstImgStride, dstPitch, W, H, t);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

// cv::GaussianBlur with 9 .. 15 taps (same fixed-point arithmetic as k_sep7: 8-bit tap fractions, one rounding at the end; the row sums are kept
// in 32 bits because independently rounded taps wider than 7 can sum to 258 or 259 -- 259 * 255 no longer fits 16 bits, and the oracle's row pass is exact,
// BORDER_REFLECT_101): LSD's blur when it reduces the image a lot (lsd_scale < 0.74) -- no configuration of the reference does, so this is a
// plain tiled kernel, not a tuned one.  64 x 16 output tile, row sums of the tile + radius rows above and below in LDS.
struct TapsWide { int r; int t[15]; };
__global__ __launch_bounds__(256) void k_sep_wide(const uint8_t* __restrict__ src, size_t srcImgStride, int srcPitch, uint8_t* __restrict__ dst,
                                                  size_t dstImgStride, int dstPitch, int W, int H, TapsWide taps)
{
    constexpr int TW = 64, TH = 16, RMAX = 7;
    __shared__ uint32_t s_h[TH + 2 * RMAX][TW];
    const int r = taps.r, n = 2 * r + 1;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const uint8_t* in = src + (size_t)blockIdx.z * srcImgStride;
    uint8_t* out = dst + (size_t)blockIdx.z * dstImgStride;
    for (int i = threadIdx.x; i < (TH + 2 * r) * TW; i += 256) {
        const int ry = i / TW, cx = i - ry * TW, x = x0 + cx;
        int yy = y0 - r + ry;
        if (yy < 0) yy = -yy;
        if (yy >= H) yy = 2 * (H - 1) - yy;
        int acc = 0;
        if (x < W && yy >= 0 && yy < H) {
            const uint8_t* row = in + (size_t)yy * srcPitch;
            for (int k = 0; k < n; ++k) {
                int xx = x + k - r;
                if (xx < 0) xx = -xx;
                if (xx >= W) xx = 2 * (W - 1) - xx;
                acc += taps.t[k] * (int)row[xx];
            }
        }
        s_h[ry][cx] = (uint32_t)acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int ty = i / TW, cx = i - ty * TW, x = x0 + cx, y = y0 + ty;
        if (x >= W || y >= H) continue;
        int acc = 0;
        for (int k = 0; k < n; ++k) acc += taps.t[k] * (int)s_h[ty + k][cx];
        const int v = (acc + 32768) >> 16;
        out[(size_t)y * dstPitch + x] = (uint8_t)(v > 255 ? 255 : v);
    }
}

int launch_sep_wide(const uint8_t* src, size_t srcImgStride, int srcPitch, uint8_t* dst, size_t dstImgStride, int dstPitch, int W, int H,
                    const int* taps, int r, int n_images, hipStream_t s)
{
    if (r < 1 || r > 7 || 2 * r >= W || 2 * r >= H) { set_error("launch_sep_wide: radius"); return OLF_ERR_INVALID; }
    TapsWide t; t.r = r;
    for (int i = 0; i < 15; ++i) t.t[i] = i < 2 * r + 1 ? taps[i] : 0;
    hipLaunchKernelGGL(k_sep_wide, dim3((W + 63) / 64, (H + 15) / 16, n_images), dim3(256), 0, s, src, srcImgStride, srcPitch, dst, dstImgStride,
                       dstPitch, W, H, t);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

// ---------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR) for 8UC1 with host-built coefficient tables, LDS tiled:
// a block produces 256 x 8 output pixels; the source rows/columns it needs are staged once with word loads,
// the horizontal pass (S[sx]*a0 + S[sx+1]*a1) >> 4 is computed once per (source row, output column) into LDS as u16
// (the reference only ever uses h >> 4, which fits: 255*2048 >> 4 = 32640), the vertical pass reads two of them.
constexpr int RZ_TW = 256, RZ_TH = 8, RZ_SW = 96 /* words per staged source row */, RZ_SH = 16;

__global__ __launch_bounds__(256) void k_resize_tiled(const uint8_t* __restrict__ src, size_t srcImgStride, int srcPitch, int sw, int sh,
                                                      uint8_t* __restrict__ dst, size_t dstImgStride, int dstPitch, int dw, int dh,
                                                      const ResizeCoef* __restrict__ rx, const ResizeCoef* __restrict__ ry)
{
    __shared__ uint32_t s_src[RZ_SH * RZ_SW];
    __shared__ uint16_t s_h[RZ_SH * RZ_TW];
    __shared__ ResizeCoef s_rx[RZ_TW];
    const int img = blockIdx.z, dx0 = blockIdx.x * RZ_TW, dy0 = blockIdx.y * RZ_TH;
    const int nx = min(RZ_TW, dw - dx0), ny = min(RZ_TH, dh - dy0);
    const uint8_t* s = src + (size_t)img * srcImgStride;
    // source window: rows [ry0, ry1], columns from the aligned word holding rx[dx0].ofs
    const int ry0 = min(max((int)ry[dy0].ofs, 0), sh - 1), ry1 = min(max((int)ry[dy0 + ny - 1].ofs + 1, 0), sh - 1);
    const int cx0 = ((int)rx[dx0].ofs) & ~3;
    const int cx1 = min((int)rx[dx0 + nx - 1].ofs + 1, sw - 1);
    const int nrows = ry1 - ry0 + 1, nwords = (cx1 - cx0) / 4 + 1;
    if (threadIdx.x < nx) s_rx[threadIdx.x] = rx[dx0 + threadIdx.x];
    for (int i = threadIdx.x; i < nrows * nwords; i += 256) {
        const int r = i / nwords, j = i - r * nwords;
        // rows are 4-byte aligned (pitch % 4 == 0 and aligned base); the last word may run past sw but stays inside the pitch
        s_src[r * RZ_SW + j] = *reinterpret_cast<const uint32_t*>(s + (size_t)(ry0 + r) * srcPitch + cx0 + 4 * j);
    }
    __syncthreads();
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(s_src);
    if ((int)threadIdx.x < nx) {
        // thread x owns output column x of the tile for every staged source row: its two source bytes and weights are fixed
        const int x = threadIdx.x;
        const ResizeCoef c = s_rx[x];
        const int o0 = (int)c.ofs - cx0, o1 = min((int)c.ofs + 1, sw - 1) - cx0;
        const int a0 = c.a0, a1 = c.a1;
        for (int r = 0; r < nrows; ++r) {
            const int h = (int)sb[r * RZ_SW * 4 + o0] * a0 + (int)sb[r * RZ_SW * 4 + o1] * a1;
            s_h[r * RZ_TW + x] = (uint16_t)(h >> 4);
        }
    }
    __syncthreads();
    uint8_t* d = dst + (size_t)img * dstImgStride;
    for (int i = threadIdx.x; i < ny * (RZ_TW / 4); i += 256) {
        const int yy = i >> 6, q = i & 63;
        if (4 * q >= nx) continue;
        const ResizeCoef cy = ry[dy0 + yy];
        const int r0 = min(max((int)cy.ofs, 0), sh - 1) - ry0, r1 = min(max((int)cy.ofs + 1, 0), sh - 1) - ry0;
        const int b0 = cy.a0, b1 = cy.a1;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h0 = s_h[r0 * RZ_TW + 4 * q + k], h1 = s_h[r1 * RZ_TW + 4 * q + k];
            const int v = (((b0 * h0) >> 16) + ((b1 * h1) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xff) << (8 * k);
        }
        // columns beyond dw inside the last quad are scratch bytes of the padded pitch
        *reinterpret_cast<uint32_t*>(d + (size_t)(dy0 + yy) * dstPitch + dx0 + 4 * q) = out;
    }
}

// ---- the same resize without LDS (round 4): one thread = 4 output columns x RS_ROWS output rows, one wave per block so that the row bookkeeping is scalar.
// The two source bytes of a column sit inside an 8-byte window that starts at the quad's first source column (host check: resize_strip_fits); a
// v_perm_b32 puts them into the halves of a dword and one v_dot2_u32_u16 with the packed (a0, a1) gives S[sx] a0 + S[sx1] a1.  A source row's sums are
// kept while the next output row still needs them (every second row of a x1.2 reduction, most rows of LSD's x1.2 enlargement).  15 instead of 40 vector
// instructions per pixel; no staging, no barriers.
constexpr int RS_ROWS = 8;

__global__ __launch_bounds__(64) void k_resize_strip(const uint8_t* __restrict__ src, size_t srcImgStride, int srcPitch, int sw, int sh,
                                                     uint8_t* __restrict__ dst, size_t dstImgStride, int dstPitch, int dw, int dh,
                                                     const ResizeCoef* __restrict__ rx, const ResizeCoef* __restrict__ ry, int nsx)
{
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= nsx) return;
    const int dx0 = 4 * q, dy0 = blockIdx.y * RS_ROWS;
    const uint8_t* s = src + (size_t)blockIdx.z * srcImgStride;
    uint8_t* d = dst + (size_t)blockIdx.z * dstImgStride + dx0;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    uint32_t sel[4], coef[4];
    const int base = min((int)rx[dx0].ofs, srcPitch - 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const ResizeCoef c = rx[min(dx0 + k, dw - 1)];
        const int o0 = (int)c.ofs - base, o1 = min((int)c.ofs + 1, sw - 1) - base;
        sel[k] = (uint32_t)o0 | 0x0c00u | ((uint32_t)o1 << 16) | 0x0c000000u;      // (byte o0, 0, byte o1, 0) of the 8-byte window
        coef[k] = (uint32_t)(uint16_t)c.a0 | ((uint32_t)(uint16_t)c.a1 << 16);
    }
    struct Raw { uint32_t w[2]; };
    auto load_row = [&](int r) -> Raw { Raw v; __builtin_memcpy(v.w, s + (size_t)r * srcPitch + base, 8); return v; };
    // (S[sx] a0 + S[sx1] a1) >> 4 of the quad's four columns from a source row's 8-byte window -- the reference only ever uses h >> 4
    auto hrow = [&](const Raw& v, uint32_t* h) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t pr = __builtin_amdgcn_perm(v.w[1], v.w[0], sel[k]);
            us2 va, vb;
            __builtin_memcpy(&va, &pr, 4); __builtin_memcpy(&vb, &coef[k], 4);
            h[k] = __builtin_amdgcn_udot2(va, vb, 0u, false) >> 4;
        }
    };
    auto rows_of = [&](int dy, int& r0, int& r1, uint32_t& b0, uint32_t& b1) {
        const ResizeCoef cy = ry[min(dy, dh - 1)];
        r0 = min(max((int)cy.ofs, 0), sh - 1); r1 = min(max((int)cy.ofs + 1, 0), sh - 1);
        b0 = (uint32_t)cy.a0; b1 = (uint32_t)cy.a1;
    };
    // (everything about rows is wave-uniform: one strip row per block.)  A source row's sums are kept while the next output row still needs them; the
    // rows an output row needs and does not have are requested one output row ahead, so their latency is covered by a row's arithmetic
    int r0, r1, ra = -1, rb = -1;
    uint32_t b0, b1;
    rows_of(dy0, r0, r1, b0, b1);
    Raw p0 = load_row(r0), p1 = load_row(r1);
    uint32_t hA[4] = {0, 0, 0, 0}, hB[4] = {0, 0, 0, 0};
#pragma unroll
    for (int rr = 0; rr < RS_ROWS; ++rr) {
        const int dy = dy0 + rr;
        if (dy >= dh) break;
        int n0, n1;
        uint32_t c0, c1;
        rows_of(dy + 1, n0, n1, c0, c1);
        Raw q0 = p0, q1 = p1;
        if (rr + 1 < RS_ROWS) {
            if (n0 != r0 && n0 != r1) q0 = load_row(n0);
            if (n1 != n0 && n1 != r1) q1 = load_row(n1);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t h0[4], h1[4];
        if (r0 == ra) { for (int k = 0; k < 4; ++k) h0[k] = hA[k]; }
        else if (r0 == rb) { for (int k = 0; k < 4; ++k) h0[k] = hB[k]; }
        else hrow(p0, h0);
        if (r1 == r0) { for (int k = 0; k < 4; ++k) h1[k] = h0[k]; }
        else if (r1 == rb) { for (int k = 0; k < 4; ++k) h1[k] = hB[k]; }
        else hrow(p1, h1);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t v = (((b0 * h0[k]) >> 16) + ((b1 * h1[k]) >> 16) + 2u) >> 2;
            out |= (v & 0xffu) << (8 * k);
        }
        // columns beyond dw inside the last quad are scratch bytes of the padded pitch
        __builtin_memcpy(d + (size_t)dy * dstPitch, &out, 4);
        ra = r0; rb = r1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { hA[k] = h0[k]; hB[k] = h1[k]; }
        r0 = n0; r1 = n1; b0 = c0; b1 = c1; p0 = q0; p1 = q1;
    }
}

// host: every quad of output columns finds its source bytes inside 8 consecutive bytes (any scale factor up to about 1.7)
bool resize_strip_fits(const ResizeCoef* rx, int sw, int dw, int ncols)
{
    for (int x0 = 0; x0 < dw; x0 += 4) {
        const int last = std::min(x0 + ncols - 1, dw - 1);
        if (std::min((int)rx[last].ofs + 1, sw - 1) - (int)rx[x0].ofs > 7) return false;
    }
    return sw >= 8;
}

// host: largest source window any tile needs; the tiled kernel is used when it fits the LDS staging area
bool resize_tiled_fits(const ResizeCoef* rx, const ResizeCoef* ry, int sw, int sh, int dw, int dh)
{
    for (int x0 = 0; x0 < dw; x0 += RZ_TW) {
        const int nx = std::min(RZ_TW, dw - x0);
        const int c0 = ((int)rx[x0].ofs) & ~3, c1 = std::min((int)rx[x0 + nx - 1].ofs + 1, sw - 1);
        if ((c1 - c0) / 4 + 1 > RZ_SW) return false;
    }
    for (int y0 = 0; y0 < dh; y0 += RZ_TH) {
        const int ny = std::min(RZ_TH, dh - y0);
        const int r0 = std::min(std::max((int)ry[y0].ofs, 0), sh - 1), r1 = std::min(std::max((int)ry[y0 + ny - 1].ofs + 1, 0), sh - 1);
        if (r1 - r0 + 1 > RZ_SH) return false;
    }
    return true;
}

int launch_resize_tiled(const uint8_t* src, size_t srcImgStride, int srcPitch, int sw, int sh, uint8_t* dst, size_t dstImgStride, int dstPitch,
                        int dw, int dh, const ResizeCoef* d_rx, const ResizeCoef* d_ry, int n_images, hipStream_t s, bool strip)
{
    static const bool stripsOn = [] { const char* e = getenv("OLF_RESIZE"); return !e || atoi(e) != 0; }();
    strip = strip && stripsOn && (dstPitch & 3) == 0 && (dstImgStride & 3) == 0 && srcPitch >= 8;
    if (strip) {      // (resize_strip_fits on the host tables; OLF_RESIZE=0 keeps the LDS-tiled kernel for A/B measurements)
        const int nsx = (dw + 3) / 4;
        hipLaunchKernelGGL(k_resize_strip, dim3((nsx + 63) / 64, (dh + RS_ROWS - 1) / RS_ROWS, n_images), dim3(64), 0, s, src, srcImgStride, srcPitch, sw, sh,
                           dst, dstImgStride, dstPitch, dw, dh, d_rx, d_ry, nsx);
        OLF_HIP_CHECK(hipGetLastError());
        return OLF_OK;
    }
    hipLaunchKernelGGL(k_resize_tiled, dim3((dw + RZ_TW - 1) / RZ_TW, (dh + RZ_TH - 1) / RZ_TH, n_images), dim3(256), 0, s, src, srcImgStride,
                       srcPitch, sw, sh, dst, dstImgStride, dstPitch, dw, dh, d_rx, d_ry);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
