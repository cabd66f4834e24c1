"""GPU checks of device arithmetic that must equal IEEE / the CPU bit for bit."""
import ctypes as C
import pytest
from orb_line_slam_amd import _lib

pytestmark = pytest.mark.gpu


def test_agent_division_is_ieee_exact():
    """The LSD agent divides with v_rcp_f32 + Newton / residual steps and no range scaling (lsd.hip fdiv_unscaled): over 2^30 operand pairs
    from its operand range (0 <= a <= b, b in [2^-45, 2^19), incl. a == b and a == 0) every quotient equals the IEEE division's."""
    ctx = _lib.Context(_lib.default_params(), 640, 480, 1)
    for seed in (1, 0xDEADBEEF, 20260928):
        bad = C.c_uint64(123)
        _lib.check(_lib.lib().olf_debug_fdiv_sweep(ctx.handle, seed, 16384, 256 // 3 + 1, C.byref(bad)), "olf_debug_fdiv_sweep")
        assert bad.value == 0, (seed, bad.value)


def test_key_kernel_sqrt_is_ieee_exact():
    """k_lsd_keys takes sqrt(n / 4.0) of n = gx^2 + gy^2 <= 2 * 510^2 with the compiler's expansion stripped of its range scaling (lsd_device.hpp
    sqrt_quarter): every n up to 2^20 must give the bits of the IEEE square root.  The same sweep checks the float-first bin (lsd.hip lsd_bin: float product,
    the double expression only within 10^-3 of an integer) against int(sqrt(n / 4.0) * bin_coef) for 97 image maxima and every n that can occur under them."""
    ctx = _lib.Context(_lib.default_params(), 640, 480, 1)
    bad = C.c_uint64(123)
    _lib.check(_lib.lib().olf_debug_sqrtq_sweep(ctx.handle, 1 << 20, C.byref(bad)), "olf_debug_sqrtq_sweep")
    assert bad.value == 0, bad.value


@pytest.mark.parametrize("ang_th", [22.5, 10.0, 45.0, 80.0])
def test_cheap_alignment_test_never_contradicts_the_reference(ang_th):
    """The growth agent decides `is this pixel aligned with the region` from dot / cross products of the float sums with the pixel's tabulated direction and
    forms the reference's region angle (cv::fastAtan2 of the sums) only for the decisions inside a margin around the tolerance (lsd.hip, PF bit 16).  Over
    3 x 2^28 (sums, candidate) pairs, three quarters of them within 3 mrad of the tolerance, no decision it calls certain differs from the reference's
    |fastAtan2(sums) * DEG2RAD - angle| <= prec, and the margin is not hit by more than the share of the samples that was aimed at it."""
    p = _lib.default_params()
    p.line.lsd_ang_th = ang_th
    ctx = _lib.Context(p, 640, 480, 1)
    for seed in (3, 0xC0FFEE, 20260929):
        out = (C.c_uint64 * 3)()
        _lib.check(_lib.lib().olf_debug_align_sweep(ctx.handle, seed, 4096, 256, out), "olf_debug_align_sweep")
        assert out[2] == 4096 * 256 * 256
        assert out[0] == 0, (ang_th, seed, list(out))
        assert out[1] < out[2] // 2, list(out)
