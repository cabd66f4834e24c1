// filters.hip -- the 8-bit fixed-point separable Gaussian shared by the path, gfx950:
//   GaussianBlur(7x7, sigma 2)   per pyramid level, reference src/ORBextractor.cc:1087-1088
//   GaussianBlur(7x7, sigma 0.6) inside cv::LineSegmentDetector (App. A.7 step 1)
//   GaussianBlur(5x5, sigma 1)   BinaryDescriptor::computeGaussianPyramid, binary_descriptor_custom.cpp:358
// Row pass exact in u16 (taps sum <= 257, 257*255 = 65535), column pass (sum + 2^15) >> 16 saturated, BORDER_REFLECT_101.
// HBM-bound: each block reads a (128+8) x (32+6) byte halo tile with 4-byte loads (v_alignbit for rows that are not
// 4-byte aligned), filters 4 pixels per thread out of LDS words and writes 4-byte words: one read + one write of the image.
#include "olf_internal.hpp"

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
    __shared__ uint32_t s_h[SF_INH * (SF_TW / 2)];             // row-pass results, two u16 per word
    const int img = blockIdx.z, x0 = blockIdx.x * SF_TW, y0 = blockIdx.y * SF_TH;
    const uint8_t* s = src + (size_t)img * srcImgStride;
    for (int i = threadIdx.x; i < SF_INH * SF_INW; i += 256) {
        const int r = i / SF_INW, j = i - r * SF_INW;
        const int gy = sf_reflect(min(y0 - 3 + r, H + 2), H);
        const int xw = x0 - 4 + 4 * j;
        const uint8_t* row = s + (size_t)gy * srcPitch;
        uint32_t v;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(row + xw);
        const int sh = (int)(addr & 3) * 8;
        if (xw >= 0 && (sh ? xw + 7 < W : xw + 3 < W)) {          // the second aligned word must stay inside the row
            const uint32_t* p4 = reinterpret_cast<const uint32_t*>(addr & ~(uintptr_t)3);
            v = sh ? __funnelshift_r(p4[0], p4[1], sh) : p4[0];
        } else {
            v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint32_t)row[sf_reflect(min(xw + b, W + 2), W)] << (8 * b);
        }
        s_in[i] = v;
    }
    __syncthreads();
    // ---- row pass: 4 outputs per task from bytes [4q+1, 4q+10] of the tile row (tile byte 4 == image x0)
    for (int i = threadIdx.x; i < SF_INH * (SF_TW / 4); i += 256) {
        const int r = i / (SF_TW / 4), q = i - r * (SF_TW / 4);
        const uint32_t w0 = s_in[r * SF_INW + q], w1 = s_in[r * SF_INW + q + 1], w2 = s_in[r * SF_INW + q + 2];
        int b[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) { b[k] = (w0 >> (8 * k)) & 0xff; b[4 + k] = (w1 >> (8 * k)) & 0xff; b[8 + k] = (w2 >> (8 * k)) & 0xff; }
        int o[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int acc = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) acc += taps.t[k] * b[p + 1 + k];    // output x = x0+4q+p uses x-3 .. x+3 = tile bytes 4q+p+1 ..
            o[p] = acc;
        }
        s_h[r * (SF_TW / 2) + 2 * q] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        s_h[r * (SF_TW / 2) + 2 * q + 1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
    }
    __syncthreads();
    // ---- column pass: each thread 4 pixels wide x 4 rows tall (10 rows of row-pass words)
    const int q = threadIdx.x & 31, ry0 = (threadIdx.x >> 5) * 4;
    const int gx = x0 + 4 * q;
    if (gx >= W) return;
    uint32_t h0[10], h1[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { h0[k] = s_h[(ry0 + k) * (SF_TW / 2) + 2 * q]; h1[k] = s_h[(ry0 + k) * (SF_TW / 2) + 2 * q + 1]; }
    uint8_t* d = dst + (size_t)img * dstImgStride;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int gy = y0 + ry0 + rr;
        if (gy >= H) break;
        int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int t = taps.t[k];
            a0 += t * (int)(h0[rr + k] & 0xffff); a1 += t * (int)(h0[rr + k] >> 16);
            a2 += t * (int)(h1[rr + k] & 0xffff); a3 += t * (int)(h1[rr + k] >> 16);
        }
        a0 = min((a0 + 32768) >> 16, 255); a1 = min((a1 + 32768) >> 16, 255);
        a2 = min((a2 + 32768) >> 16, 255); a3 = min((a3 + 32768) >> 16, 255);
        uint8_t* o = d + (size_t)gy * dstPitch + gx;
        if (gx + 3 < W && ((reinterpret_cast<uintptr_t>(o) & 3) == 0))
            *reinterpret_cast<uint32_t*>(o) = (uint32_t)a0 | ((uint32_t)a1 << 8) | ((uint32_t)a2 << 16) | ((uint32_t)a3 << 24);
        else {
            o[0] = (uint8_t)a0;
            if (gx + 1 < W) o[1] = (uint8_t)a1;
            if (gx + 2 < W) o[2] = (uint8_t)a2;
            if (gx + 3 < W) o[3] = (uint8_t)a3;
        }
    }
}

int launch_sep7(const uint8_t* src, size_t srcImgStride, int srcPitch, uint8_t* dst, size_t dstImgStride, int dstPitch, int W, int H,
                const int* taps7, int n_images, hipStream_t s)
{
    Taps7 t;
    for (int i = 0; i < 7; ++i) t.t[i] = taps7[i];
    hipLaunchKernelGGL(k_sep7, dim3((W + SF_TW - 1) / SF_TW, (H + SF_TH - 1) / SF_TH, n_images), dim3(256), 0, s, src, srcImgStride, srcPitch,
                       dst, dstImgStride, dstPitch, W, H, t);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
