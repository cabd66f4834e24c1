// precond.hip -- input conditioning ahead of the feature path, gfx950:
//   cvtColor RGB/BGR(A) -> GRAY        reference src/Tracking.cc:193-218 (Tracking::GrabImageStereo)
//   remap(INTER_LINEAR) rectification  reference Examples/PL/PL_stereo_euroc.cc:136-137 (maps from initUndistortRectifyMap,
//                                      computed once at start-up on the host, :97-98)
// Pure per-pixel HBM-bound kernels: keeping them on the device removes a host pass and the H2D of un-rectified data.
// OpenCV 8-bit fixed-point semantics (restated in oracle/precond_oracle.cpp).
#include "olf_internal.hpp"

namespace olf {

__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int npx, int cn, int bgr)
{
    const size_t img = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= npx) return;
    const uint8_t* s = src + (img * npx + i0) * cn;
    uint8_t* d = dst + img * npx + i0;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < npx) {
            const uint8_t* p = s + k * cn;
            const int r = bgr ? p[2] : p[0], g = p[1], b = bgr ? p[0] : p[2];
            out |= (uint32_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14) << (8 * k);
        }
    }
    if (i0 + 3 < npx && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) *reinterpret_cast<uint32_t*>(d) = out;
    else
        for (int k = 0; k < 4 && i0 + k < npx; ++k) d[k] = (uint8_t)(out >> (8 * k));
}

__global__ __launch_bounds__(256) void k_remap_linear(const uint8_t* __restrict__ src, int sw, int sh, const float* __restrict__ mapx,
                                                      const float* __restrict__ mapy, int dw, int dh, uint8_t* __restrict__ dst)
{
    const size_t img = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dw * dh) return;
    const uint8_t* s = src + img * (size_t)sw * sh;
    const int sxf = __float2int_rn(__fmul_rn(mapx[i], 32.f)), syf = __float2int_rn(__fmul_rn(mapy[i], 32.f));
    const int fx = sxf & 31, fy = syf & 31;
    const int sx = min(max(sxf >> 5, -32768), 32767), sy = min(max(syf >> 5, -32768), 32767);
    const int w0 = 32 * (32 - fy) * (32 - fx), w1 = 32 * (32 - fy) * fx, w2 = 32 * fy * (32 - fx), w3 = 32 * fy * fx;
    const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw, y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
    const int p00 = (x0 && y0) ? s[(size_t)sy * sw + sx] : 0, p01 = (x1 && y0) ? s[(size_t)sy * sw + sx + 1] : 0;
    const int p10 = (x0 && y1) ? s[(size_t)(sy + 1) * sw + sx] : 0, p11 = (x1 && y1) ? s[(size_t)(sy + 1) * sw + sx + 1] : 0;
    const int v = (p00 * w0 + p01 * w1 + p10 * w2 + p11 * w3 + (1 << 14)) >> 15;
    dst[img * (size_t)dw * dh + i] = (uint8_t)min(max(v, 0), 255);
}

int launch_cvt_gray(const uint8_t* src, uint8_t* dst, int w, int h, int code, int n_images, hipStream_t s)
{
    const int npx = w * h;
    hipLaunchKernelGGL(k_cvt_gray, dim3((npx / 4 + 256) / 256, n_images), dim3(256), 0, s, src, dst, npx, code >= 2 ? 4 : 3, code & 1);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_remap_linear(const uint8_t* src, int sw, int sh, const float* mapx, const float* mapy, int dw, int dh, uint8_t* dst, int n_images,
                        hipStream_t s)
{
    hipLaunchKernelGGL(k_remap_linear, dim3((dw * dh + 255) / 256, n_images), dim3(256), 0, s, src, sw, sh, mapx, mapy, dw, dh, dst);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
