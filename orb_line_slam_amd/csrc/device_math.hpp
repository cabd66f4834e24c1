// device_math.hpp -- scalar float routines whose results must equal the reference's host libm /
// OpenCV results bit for bit.  Compiles for the device (hipcc) and for the host (g++, used by the
// CPU test that sweeps these against glibc), always without FMA contraction.
//
//  * dev_fastAtan2  : cv::fastAtan2 -> atanImpl<float> (OpenCV 3.4 mathfuncs)
//  * glibc_cosf/sinf: the reference calls std::cos(float)/std::sin(float) in
//                     src/ORBextractor.cc:114-115 (namespace std is imported at :69), i.e. glibc's
//                     cosf/sinf.  glibc >= 2.28 implements them with the ARM "optimized routines"
//                     algorithm (sysdeps/ieee754/flt-32/s_cosf.c, sincosf.h): double-precision
//                     argument reduction by pi/2 and two fixed polynomials.  The routine below is
//                     that published algorithm with the coefficient table as shipped in glibc 2.35
//                     (read from libm.so.6's __sincosf_table); only IEEE double mul/add/sub and one
//                     double->float rounding are involved, so device and host agree exactly.
//                     tests/test_device_math.py sweeps it against the box's libm.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define OLF_HD __host__ __device__ __forceinline__
#else
#define OLF_HD static inline
#endif

namespace olf {

#if defined(__HIP_DEVICE_COMPILE__)
OLF_HD float f_mul(float a, float b) { return __fmul_rn(a, b); }
OLF_HD float f_add(float a, float b) { return __fadd_rn(a, b); }
OLF_HD float f_sub(float a, float b) { return __fsub_rn(a, b); }
OLF_HD float f_div(float a, float b) { return __fdiv_rn(a, b); }
OLF_HD double d_mul(double a, double b) { return __dmul_rn(a, b); }
OLF_HD double d_add(double a, double b) { return __dadd_rn(a, b); }
OLF_HD double d_sub(double a, double b) { return __dsub_rn(a, b); }
#else   // host: the translation unit is compiled with -ffp-contract=off
OLF_HD float f_mul(float a, float b) { return a * b; }
OLF_HD float f_add(float a, float b) { return a + b; }
OLF_HD float f_sub(float a, float b) { return a - b; }
OLF_HD float f_div(float a, float b) { return a / b; }
OLF_HD double d_mul(double a, double b) { return a * b; }
OLF_HD double d_add(double a, double b) { return a + b; }
OLF_HD double d_sub(double a, double b) { return a - b; }
#endif

OLF_HD float dev_fastAtan2(float y, float x)
{
    // the four coefficients are float products evaluated in float, as in OpenCV's static initialisers
    const float k = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;   // (float)DBL_EPSILON
    const float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    float a, c, c2;
    if (ax >= ay) {
        c = f_div(ay, f_add(ax, eps));
        c2 = f_mul(c, c);
        a = f_mul(f_add(f_mul(f_add(f_mul(f_add(f_mul(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = f_div(ax, f_add(ay, eps));
        c2 = f_mul(c, c);
        a = f_sub(90.f, f_mul(f_add(f_mul(f_add(f_mul(f_add(f_mul(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = f_sub(180.f, a);
    if (y < 0) a = f_sub(360.f, a);
    return a;
}

// ---- glibc 2.35 sincosf ----------------------------------------------------------------------
struct sincosf_tab {
    double sign[4];
    double hpi_inv, hpi;
    double c0, c1, s1, c2, s2, c3, s3, c4;   // memory order of __sincosf_table
};

OLF_HD double bits2d(uint64_t u)
{
    double d;
#if defined(__HIP_DEVICE_COMPILE__)
    d = __longlong_as_double((long long)u);
#else
    memcpy(&d, &u, 8);
#endif
    return d;
}

// evaluates the polynomial pair; q selects the table entry (0: +cos coefficients, 1: negated)
OLF_HD float sincosf_poly(double x, double x2, int q, int n)
{
    const double s1 = bits2d(0xBFC555545995A603ULL);                    // -0x1.555545995a603p-3
    const double s2 = bits2d(0x3F81107605230BC4ULL);                    //  0x1.1107605230bc4p-7
    const double s3 = bits2d(0xBF2994EB3774CF24ULL);                    // -0x1.994eb3774cf24p-13
    double c0 = 1.0;
    double c1 = bits2d(0xBFDFFFFFFD0C621CULL);                          // -0x1.ffffffd0c621cp-2
    double c2 = bits2d(0x3FA55553E1068F19ULL);                          //  0x1.55553e1068f19p-5
    double c3 = bits2d(0xBF56C087E89A359DULL);                          // -0x1.6c087e89a359dp-10
    double c4 = bits2d(0x3EF99343027BF8C3ULL);                          //  0x1.99343027bf8c3p-16
    if (q) { c0 = -c0; c1 = -c1; c2 = -c2; c3 = -c3; c4 = -c4; }
    if ((n & 1) == 0) {
        const double x3 = d_mul(x, x2);
        const double t1 = d_add(s2, d_mul(x2, s3));
        const double x7 = d_mul(x3, x2);
        const double s = d_add(x, d_mul(x3, s1));
        return (float)d_add(s, d_mul(x7, t1));
    } else {
        const double x4 = d_mul(x2, x2);
        const double t2 = d_add(c3, d_mul(x2, c4));
        const double t1 = d_add(c0, d_mul(x2, c1));
        const double x6 = d_mul(x4, x2);
        const double c = d_add(t1, d_mul(x4, c2));
        return (float)d_add(c, d_mul(x6, t2));
    }
}

OLF_HD uint32_t abstop12(float x)
{
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(x);
#else
    memcpy(&u, &x, 4);
#endif
    return (u >> 20) & 0x7ff;
}

// valid for |y| < 120 (the path only evaluates y in [0, 2*pi)); larger arguments are not needed
OLF_HD float glibc_sincosf_core(float y, int want_cos)
{
    double x = (double)y;
    if (abstop12(y) < 0x3f4) {               // |y| < pi/4   (abstop12(0x1.921FB6p-1f) == 0x3f4)
        const double x2 = d_mul(x, x);
        if (abstop12(y) < 0x398) return want_cos ? 1.0f : y;   // |y| < 2^-12
        return sincosf_poly(x, x2, 0, want_cos);
    }
    const double hpi_inv = bits2d(0x41645F306DC9C883ULL);   // 0x1.45F306DC9C883p+23  (2/pi * 2^24)
    const double hpi = bits2d(0x3FF921FB54442D18ULL);       // 0x1.921FB54442D18p0
    const double r = d_mul(x, hpi_inv);
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = d_sub(x, d_mul((double)n, hpi));
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign[] = {1,-1,-1,1}
    const int q = (n & 2) ? 1 : 0;
    return sincosf_poly(d_mul(x, sgn), d_mul(x, x), q, want_cos ? (n ^ 1) : n);
}

OLF_HD float glibc_cosf(float y) { return glibc_sincosf_core(y, 1); }
OLF_HD float glibc_sinf(float y) { return glibc_sincosf_core(y, 0); }

// ---- glibc's atan2f (2.35: the fdlibm float code, sysdeps/ieee754/flt-32/e_atan2f.c + s_atanf.c): only float mul / add / sub / div, no fused
// operations (x86-64 glibc carries no FMA variant of it), so a restatement with the same constants and the same operation order gives the same bits.
// Convention C.6, variant conv_libm_float = 1 (KeyLine.angle = atan2(dy, dx) on floats resolving to the float overload).  Arguments here are finite
// and not both zero-or-huge, but every branch of the original is kept.  tests/test_host_cpu.py sweeps it against this box's libm.
OLF_HD float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
OLF_HD uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

OLF_HD float glibc_atanf(float x)
{
    // (the decimal literals of the source, converted double -> float as the C compiler does; the hexadecimal comments of fdlibm are not all exact)
    const float atanhi0 = (float)4.6364760399e-01, atanhi1 = (float)7.8539812565e-01, atanhi2 = (float)9.8279368877e-01, atanhi3 = (float)1.5707962513e+00;
    const float atanlo0 = (float)5.0121582440e-09, atanlo1 = (float)3.7748947079e-08, atanlo2 = (float)3.4473217170e-08, atanlo3 = (float)7.5497894159e-08;
    const float aT0 = (float)3.3333334327e-01, aT1 = (float)-2.0000000298e-01, aT2 = (float)1.4285714924e-01, aT3 = (float)-1.1111110449e-01,
                aT4 = (float)9.0908870101e-02, aT5 = (float)-7.6918758452e-02, aT6 = (float)6.6610731184e-02, aT7 = (float)-5.8335702866e-02,
                aT8 = (float)4.9768779427e-02, aT9 = (float)-3.6531571299e-02, aT10 = (float)1.6285819933e-02;
    const int32_t hx = (int32_t)f2bits(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    float hi = 0.f, lo = 0.f;
    if (ix >= 0x4c000000) {                          // |x| >= 2^25
        if (ix > 0x7f800000) return f_add(x, x);     // NaN
        if (hx > 0) return f_add(atanhi3, atanlo3);
        return f_sub(-atanhi3, atanlo3);
    }
    if (ix < 0x3ee00000) {                           // |x| < 0.4375
        if (ix < 0x31000000) return x;               // |x| < 2^-29
        id = -1;
    } else {
        x = bits2f((uint32_t)ix);                    // fabsf
        if (ix < 0x3f980000) {                       // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; hi = atanhi0; lo = atanlo0; x = f_div(f_sub(f_mul(2.0f, x), 1.0f), f_add(2.0f, x)); }      // 7/16 <= |x| < 11/16
            else { id = 1; hi = atanhi1; lo = atanlo1; x = f_div(f_sub(x, 1.0f), f_add(x, 1.0f)); }                                    // 11/16 <= |x| < 19/16
        } else {
            if (ix < 0x401c0000) { id = 2; hi = atanhi2; lo = atanlo2; x = f_div(f_sub(x, 1.5f), f_add(1.0f, f_mul(1.5f, x))); }       // |x| < 2.4375
            else { id = 3; hi = atanhi3; lo = atanlo3; x = f_div(-1.0f, x); }                                                           // 2.4375 <= |x| < 2^25
        }
    }
    const float z = f_mul(x, x), w = f_mul(z, z);
    const float s1 = f_mul(z, f_add(aT0, f_mul(w, f_add(aT2, f_mul(w, f_add(aT4, f_mul(w, f_add(aT6, f_mul(w, f_add(aT8, f_mul(w, aT10)))))))))));
    const float s2 = f_mul(w, f_add(aT1, f_mul(w, f_add(aT3, f_mul(w, f_add(aT5, f_mul(w, f_add(aT7, f_mul(w, aT9)))))))));
    if (id < 0) return f_sub(x, f_mul(x, f_add(s1, s2)));
    const float r = f_sub(hi, f_sub(f_sub(f_mul(x, f_add(s1, s2)), lo), x));
    return hx < 0 ? -r : r;
}

OLF_HD float glibc_atan2f(float y, float x)
{
    const float tiny = (float)1.0e-30, pi_o_4 = (float)7.8539818525e-01, pi_o_2 = (float)1.5707963705e+00, pi = (float)3.1415927410e+00, pi_lo = (float)-8.7422776573e-08;
    const int32_t hx = (int32_t)f2bits(x), hy = (int32_t)f2bits(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return f_add(x, y);            // NaN
    if (hx == 0x3f800000) return glibc_atanf(y);                             // x == 1.0
    const int m = (int)(((uint32_t)hy >> 31) & 1u) | (int)(((uint32_t)hx >> 30) & 2u);      // 2 * sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
            case 0: case 1: return y;
            case 2: return f_add(pi, tiny);
            default: return f_sub(-pi, tiny);
        }
    }
    if (ix == 0) return hy < 0 ? f_sub(-pi_o_2, tiny) : f_add(pi_o_2, tiny);
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return f_add(pi_o_4, tiny);
                case 1: return f_sub(-pi_o_4, tiny);
                case 2: return f_add(f_mul(3.0f, pi_o_4), tiny);
                default: return f_sub(f_mul(-3.0f, pi_o_4), tiny);
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return f_add(pi, tiny);
                default: return f_sub(-pi, tiny);
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? f_sub(-pi_o_2, tiny) : f_add(pi_o_2, tiny);
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = f_add(pi_o_2, f_mul(0.5f, pi_lo));
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = glibc_atanf(bits2f(f2bits(f_div(y, x)) & 0x7fffffffu));
    switch (m) {
        case 0: return z;
        case 1: return bits2f(f2bits(z) ^ 0x80000000u);
        case 2: return f_sub(pi, f_sub(z, pi_lo));
        default: return f_sub(f_sub(z, pi_lo), pi);
    }
}

}  // namespace olf
