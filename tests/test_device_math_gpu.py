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
