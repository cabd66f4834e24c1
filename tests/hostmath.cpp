// Host build of orb_line_slam_amd/csrc/device_math.hpp (the same source the kernels compile) so the CPU suite
// can sweep the device routines against this box's libm / the oracle.  Compile with -ffp-contract=off.
#include "../orb_line_slam_amd/csrc/device_math.hpp"
#include <math.h>
extern "C" {
float hm_cosf(float x) { return olf::glibc_cosf(x); }
float hm_sinf(float x) { return olf::glibc_sinf(x); }
float hm_fast_atan2(float y, float x) { return olf::dev_fastAtan2(y, x); }
float hm_atan2f(float y, float x) { return olf::glibc_atan2f(y, x); }
// glibc_atan2f against libm's atan2f: a pseudo-random stream of `n` float pairs (seeded; mantissas random, exponents from the range KeyLine end-point
// differences take, plus zeros, equal operands and sign combinations) and the small-integer grid; returns the number of bit mismatches
// the same sweep over the NEGATIVE floats [-hi, -lo] (kl.angle = atan2 lies in [-pi, pi])
long hm_sweep_sincos_neg(float lo, float hi, unsigned step)
{
    unsigned a, b; memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
    long bad = 0;
    for (unsigned long u = a; u <= b; u += step) {
        unsigned v = (unsigned)u | 0x80000000u; float x; memcpy(&x, &v, 4);
        float c = olf::glibc_cosf(x), s = olf::glibc_sinf(x), rc = cosf(x), rs = sinf(x);
        if (memcmp(&c, &rc, 4) || memcmp(&s, &rs, 4)) ++bad;
    }
    return bad;
}
long hm_sweep_atan2f(unsigned long seed, long n)
{
    long bad = 0;
    unsigned long long st = seed * 0x9E3779B97F4A7C15ull + 1;
    for (long i = 0; i < n; ++i) {
        st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
        const unsigned long long r = st * 0x2545F4914F6CDD1Dull;
        float y = ldexpf(1.0f + (float)((r >> 8) & 0x7fffff) * 1.1920929e-7f, (int)(r & 31) - 12);
        float x = ldexpf(1.0f + (float)((r >> 36) & 0x7fffff) * 1.1920929e-7f, (int)((r >> 60) & 15) + (int)((r >> 32) & 15) - 12);
        if (r & (1ull << 5)) y = -y;
        if (r & (1ull << 6)) x = -x;
        if (((r >> 61) & 7) == 0) x = y;
        if (((r >> 58) & 7) == 0) y = 0.f;
        if (((r >> 55) & 7) == 0) x = 0.f;
        if (((r >> 52) & 15) == 0) x = 1.0f;
        const float a = olf::glibc_atan2f(y, x), b = atan2f(y, x);
        if (memcmp(&a, &b, 4)) ++bad;
    }
    for (int yi = -600; yi <= 600; ++yi)
        for (int xi = -600; xi <= 600; xi += 3) {
            const float a = olf::glibc_atan2f((float)yi * 0.25f, (float)xi * 0.5f), b = atan2f((float)yi * 0.25f, (float)xi * 0.5f);
            if (memcmp(&a, &b, 4)) ++bad;
        }
    return bad;
}
// sweeps every `step`-th float in [lo, hi]; returns the number of bit mismatches against libm's cosf / sinf
long hm_sweep_sincos(float lo, float hi, unsigned step)
{
    unsigned a, b; memcpy(&a, &lo, 4); memcpy(&b, &hi, 4);
    long bad = 0;
    for (unsigned long u = a; u <= b; u += step) {
        unsigned v = (unsigned)u; float x; memcpy(&x, &v, 4);
        float c = olf::glibc_cosf(x), s = olf::glibc_sinf(x), rc = cosf(x), rs = sinf(x);
        if (memcmp(&c, &rc, 4) || memcmp(&s, &rs, 4)) ++bad;
    }
    return bad;
}
}
