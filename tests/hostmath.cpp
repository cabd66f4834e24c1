// Host build of orb_line_slam_amd/csrc/device_math.hpp (the same source the kernels compile) so the CPU suite
// can sweep the device routines against this box's libm / the oracle.  Compile with -ffp-contract=off.
#include "../orb_line_slam_amd/csrc/device_math.hpp"
#include <math.h>
extern "C" {
float hm_cosf(float x) { return olf::glibc_cosf(x); }
float hm_sinf(float x) { return olf::glibc_sinf(x); }
float hm_fast_atan2(float y, float x) { return olf::dev_fastAtan2(y, x); }
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
