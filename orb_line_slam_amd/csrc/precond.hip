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

// cv::initUndistortRectifyMap, CV_32FC1 maps (Examples/PL/PL_stereo_euroc.cc:97-98): one thread per image ROW, because the source coordinates
// are a running double sum along the row in OpenCV's code (x += ir[0] per column) -- the maps are computed once per sequence, so the row
// loop costs nothing that matters, and it keeps the sums in the reference's order.  No FMA contraction, IEEE divisions.
struct RectifyPrm { double ir[9]; double k[8]; double fx, fy, u0, v0; };
__global__ __launch_bounds__(64) void k_init_rectify_map(RectifyPrm q, int w, int h, float* __restrict__ map1, float* __restrict__ map2)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= h) return;
    const double k1 = q.k[0], k2 = q.k[1], p1 = q.k[2], p2 = q.k[3], k3 = q.k[4], k4 = q.k[5], k5 = q.k[6], k6 = q.k[7];
    double _x = __dadd_rn(__dmul_rn((double)i, q.ir[1]), q.ir[2]), _y = __dadd_rn(__dmul_rn((double)i, q.ir[4]), q.ir[5]),
           _w = __dadd_rn(__dmul_rn((double)i, q.ir[7]), q.ir[8]);
    for (int j = 0; j < w; ++j) {
        const double ww = __ddiv_rn(1., _w), x = __dmul_rn(_x, ww), y = __dmul_rn(_y, ww);
        const double x2 = __dmul_rn(x, x), y2 = __dmul_rn(y, y);
        const double r2 = __dadd_rn(x2, y2), _2xy = __dmul_rn(__dmul_rn(2., x), y);
        const double num = __dadd_rn(1., __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k3, r2), k2), r2), k1), r2));
        const double den = __dadd_rn(1., __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k6, r2), k5), r2), k4), r2));
        const double kr = __ddiv_rn(num, den);
        const double xd = __dadd_rn(__dadd_rn(__dmul_rn(x, kr), __dmul_rn(p1, _2xy)), __dmul_rn(p2, __dadd_rn(r2, __dmul_rn(2., x2))));
        const double yd = __dadd_rn(__dadd_rn(__dmul_rn(y, kr), __dmul_rn(p1, __dadd_rn(r2, __dmul_rn(2., y2)))), __dmul_rn(p2, _2xy));
        map1[(size_t)i * w + j] = (float)__dadd_rn(__dmul_rn(q.fx, xd), q.u0);
        map2[(size_t)i * w + j] = (float)__dadd_rn(__dmul_rn(q.fy, yd), q.v0);
        _x = __dadd_rn(_x, q.ir[0]); _y = __dadd_rn(_y, q.ir[3]); _w = __dadd_rn(_w, q.ir[6]);
    }
}

int launch_init_rectify_map(const double* ir9, const double* k8, double fx, double fy, double u0, double v0, int w, int h, float* d_map1, float* d_map2,
                            hipStream_t s)
{
    RectifyPrm q;
    for (int i = 0; i < 9; ++i) q.ir[i] = ir9[i];
    for (int i = 0; i < 8; ++i) q.k[i] = k8[i];
    q.fx = fx; q.fy = fy; q.u0 = u0; q.v0 = v0;
    hipLaunchKernelGGL(k_init_rectify_map, dim3((h + 63) / 64), dim3(64), 0, s, q, w, h, d_map1, d_map2);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
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
