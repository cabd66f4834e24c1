// lsd_device.hpp -- device helpers shared by the LSD kernels (lsd.hip: front half, rectangle fit; lsd_grow.hip: region growing).
#pragma once
#include "line_internal.hpp"
#include "device_math.hpp"

namespace olf {

constexpr double kPI = 3.1415926535897932384626433832795;
constexpr double kDegToRads = kPI / 180;
constexpr double kM32PI = (3 * kPI) / 2, kM2PI = 2 * kPI;
// grad word: bits 0-10 gx, 11-21 gy (11-bit two's complement, |g| <= 510), bit 30 NOTDEF, bit 31 USED
constexpr unsigned kNotDef = 0x40000000u, kUsed = 0x80000000u, kIso = 0x00400000u;   // bit 22: no neighbour is aligned with this pixel
__device__ __forceinline__ int unpack_gx(uint32_t p) { return ((int)(p << 21)) >> 21; }
__device__ __forceinline__ int unpack_gy(uint32_t p) { return ((int)(p << 10)) >> 21; }

// one grown region that is large enough to be fitted: its pixels are region[start .. start + n) in growth order
struct RegionRec { int start, n; double angle; };
// One entry of the per-context angle table (index: the packed gradient pair gx:11 | gy:11 of a grad word), 32 bytes = one HBM sector:
// everything region_grow needs from a pixel.  ang = fastAtan2(gx, -gy) * DEG2RAD as a double; cs / sn = cos / sin of the float-rounded
// angle (what an added pixel contributes to the sums); seed = (float)cos / (float)sin of the double angle (the sums a region starts with).
struct AngEnt { double cs, sn, ang; float2 seed; };
static_assert(sizeof(AngEnt) == 32, "AngEnt is one 32-byte sector");
__device__ __forceinline__ uint32_t pack_g(int gx, int gy) { return ((uint32_t)gx & 0x7ffu) | (((uint32_t)gy & 0x7ffu) << 11); }

__device__ __forceinline__ double shfl_d(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, l); hi = __shfl(hi, l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int rlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double rlane_d(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// a / b for 0 <= a <= b, b in the normal range and far from overflow (the agent's sums): v_rcp_f32 refined by two Newton steps on the
// reciprocal and two residual corrections on the quotient -- the core of the IEEE division expansion (correctly rounded for any
// reciprocal seed within 1 ulp) without the range scaling (v_div_scale / v_div_fixup) that these operands never need.
// tests/test_device_math_gpu.py compares it with the compiler's IEEE division on 2^30 operand pairs.
__device__ __forceinline__ float fdiv_unscaled(float a, float b)
{
    float y = __builtin_amdgcn_rcpf(b);
    y = __fmaf_rn(__fmaf_rn(-b, y, 1.0f), y, y);
    float q = __fmul_rn(a, y);
    q = __fmaf_rn(__fmaf_rn(-b, q, a), y, q);
    q = __fmaf_rn(__fmaf_rn(-b, q, a), y, q);
    return q;
}

// sqrt(n / 4.0) -- the gradient norm of ll_angle -- for an integer 0 <= n < 2^21, correctly rounded: the compiler's own expansion of sqrt(double)
// (v_rsq_f64, one Goldschmidt step, two residual corrections) without the range scaling and the special-value selects that these operands never need
// (22 -> 12 instructions).  tests/test_device_math_gpu.py compares it with sqrt() on every n the gradient can take.
__device__ __forceinline__ double sqrt_quarter(int n)
{
    const double x = (double)n * 0.25;
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __fma_rn(-h, g, 0.5);
    g = __fma_rn(g, r, g); h = __fma_rn(h, r, h);
    double d = __fma_rn(-g, g, x);
    g = __fma_rn(d, h, g);
    d = __fma_rn(-g, g, x);
    g = __fma_rn(d, h, g);
    return n ? g : 0.0;
}

// cv::fastAtan2 as dev_fastAtan2 (device_math.hpp), for the agent's region angle: |x| via source modifiers and the unscaled division.
// Only the sign of a zero result can differ from dev_fastAtan2 (x or y == -0.0f), and the region angle is only ever compared.
__device__ __forceinline__ float agent_fastAtan2(float y, float x)
{
    const float k = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    const float mn = fminf(ax, ay), mx = fmaxf(ax, ay);
    const float c = fdiv_unscaled(mn, f_add(mx, eps));
    const float c2 = f_mul(c, c);
    float a = f_mul(f_add(f_mul(f_add(f_mul(f_add(f_mul(p7, c2), p5), c2), p3), c2), p1), c);
    if (!(ax >= ay)) a = f_sub(90.f, a);
    if (x < 0) a = f_sub(180.f, a);
    if (y < 0) a = f_sub(360.f, a);
    return a;
}

}  // namespace olf
