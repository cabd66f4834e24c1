"""GPU parity for lsd_refine = 1 (cv::LSD_REFINE_STD, src/LineExtractor.cc:45 passes Config's lsd_refine through): the density check, the second
growth under the tolerance tau and reduce_region_radius run inside the growth agent; key lines and LBD descriptors must equal the oracle's
(oracle/line_oracle.cpp, convention C.14).  lsd_refine = 2 (LSD_REFINE_ADV) adds the NFA stage."""
import ctypes as C
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
from test_line_gpu import _cmp_keylines, _pattern

pytestmark = pytest.mark.gpu


def _params(oracle, nf, nl, seed_order, **kw):
    p = oracle.full_params(nf, nl, *kw.get("cam", ()))
    p.line.lsd_refine = 1
    p.line.conv_seed_order = seed_order
    return p


@pytest.mark.parametrize("w,h,nl,seed_order", [(640, 480, 200, 1), (1242, 375, 500, 1), (752, 480, 0, 1), (640, 480, 0, 0), (320, 240, 0, 1)])
def test_line_extract_refine_std(oracle, w, h, nl, seed_order):
    p = _params(oracle, 2000, nl, seed_order)
    p0 = oracle.full_params(2000, nl); p0.line.conv_seed_order = seed_order
    ex = ola.Lineextractor(nl, 0.025, lsd_refine=1, conv_seed_order=seed_order)
    rng = np.random.default_rng(w + nl)
    changed = 0
    for seed in (3, 4):
        left, right = synth.stereo_pair(seed, w, h)
        noisy = np.clip(left.astype(np.int32) + rng.integers(-12, 13, left.shape), 0, 255).astype(np.uint8)
        for img in (left, right, noisy):
            gk, gd = ex(img)
            o = oracle.line_extract(img, p.line)
            _cmp_keylines(gk, o["kls"])
            assert np.array_equal(gd, o["desc"])
            o0 = oracle.line_extract(img, p0.line)
            changed += len(o0["kls"]) != len(o["kls"]) or not np.array_equal(o0["kls"], o["kls"])
    assert changed > 0, "the refinement must change something on these images, or the test tests nothing"


@pytest.mark.parametrize("name", ["tri_x", "tri_diag", "rings", "checker", "soft_edges"])
def test_refine_std_structured_images(oracle, name):
    """large regions with drifting angles (rings: low density -> second growth and radius reduction), massive ties, pure noise"""
    w, h = 640, 360
    img = _pattern(name, w, h)
    p = _params(oracle, 1000, 0, 1)
    ex = ola.Lineextractor(0, 0.025, lsd_refine=1)
    k, d = ex(img)
    o = oracle.line_extract(img, p.line)
    _cmp_keylines(k, o["kls"])
    assert np.array_equal(d, o["desc"])
    noise = np.random.default_rng(5).integers(0, 256, (h, w), dtype=np.uint8)
    k, d = ex(noise)
    o = oracle.line_extract(noise, p.line)
    _cmp_keylines(k, o["kls"])
    assert np.array_equal(d, o["desc"])


def test_refine_std_density_threshold_and_batch(oracle):
    """another density threshold (everything is refined at 0.95, nothing at 0.05), several images per call"""
    w, h = 640, 480
    imgs = synth.stereo_batch(50, 3, w, h)
    for dth in (0.95, 0.05, 0.7):
        p = _params(oracle, 1000, 300, 1)
        p.line.lsd_density_th = dth
        ex = ola.Lineextractor(300, 0.025, lsd_refine=1, lsd_density_th=dth, max_images=len(imgs))
        kls, desc, counts = ex.extract_batch(imgs)
        for i, img in enumerate(imgs):
            o = oracle.line_extract(img, p.line)
            c = int(counts[i])
            _cmp_keylines(kls[i, :c], o["kls"])
            assert np.array_equal(desc[i, :c], o["desc"]), (dth, i)


def test_refine_std_through_the_fused_stereo_entry(oracle):
    w, h = 1242, 375
    p = oracle.full_params(2000, 500, 718.856, 386.1448)
    p.line.lsd_refine = 1
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=2)
    imgs = synth.stereo_batch(61, 2, w, h)
    f = fe.frames(imgs)
    for i in range(2):
        g = f.pair(i)
        ol, orr = oracle.line_extract(imgs[2 * i], p.line), oracle.line_extract(imgs[2 * i + 1], p.line)
        _cmp_keylines(g["mvKeys_Line"], ol["kls"])
        _cmp_keylines(g["mvKeysRight_Line"], orr["kls"])
        assert np.array_equal(g["mDescriptors_Line"], ol["desc"]) and np.array_equal(g["mDescriptorsRight_Line"], orr["desc"])
        m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
        assert np.array_equal(g["line_matches_12"], m) and np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32))


@pytest.mark.parametrize("w,h,seed_order", [(640, 480, 1), (1242, 375, 1), (752, 480, 0), (320, 240, 1)])
def test_line_extract_refine_adv(oracle, w, h, seed_order):
    """lsd_refine = 2 (LSD_REFINE_ADV): rect_improve / rect_nfa / nfa on top of STD.  The numbers of false alarms go through the device's libm
    (log, exp, pow, sinh, log10), the oracle's through glibc's; they are only compared with each other and with log_eps, so the key lines agree
    unless such a comparison is closer than the libraries' last-bit differences -- none is on these images."""
    p = oracle.full_params(2000, 0)
    p.line.lsd_refine = 2; p.line.conv_seed_order = seed_order
    p1 = oracle.full_params(2000, 0)
    p1.line.lsd_refine = 1; p1.line.conv_seed_order = seed_order
    ex = ola.Lineextractor(0, 0.025, lsd_refine=2, conv_seed_order=seed_order)
    rng = np.random.default_rng(w)
    dropped = 0
    for seed in (3, 4):
        left, right = synth.stereo_pair(seed, w, h)
        noisy = np.clip(left.astype(np.int32) + rng.integers(-12, 13, left.shape), 0, 255).astype(np.uint8)
        for img in (left, right, noisy):
            gk, gd = ex(img)
            o = oracle.line_extract(img, p.line)
            _cmp_keylines(gk, o["kls"])
            assert np.array_equal(gd, o["desc"])
            dropped += len(oracle.line_extract(img, p1.line)["kls"]) - len(o["kls"])
    assert dropped > 0, "the NFA stage must reject something on these images"


def test_refine_adv_other_thresholds_and_patterns(oracle):
    w, h = 640, 360
    for name, eps in (("rings", 0.0), ("soft_edges", 1.0), ("tri_diag", -2.0), ("checker", 0.0)):
        img = _pattern(name, w, h)
        p = oracle.full_params(1000, 0)
        p.line.lsd_refine = 2; p.line.lsd_log_eps = eps
        ex = ola.Lineextractor(0, 0.025, lsd_refine=2, lsd_log_eps=eps)
        k, d = ex(img)
        o = oracle.line_extract(img, p.line)
        _cmp_keylines(k, o["kls"])
        assert np.array_equal(d, o["desc"]), name


def test_refine_out_of_range_is_refused():
    p = _lib.default_params()
    p.line.lsd_refine = 3
    h = C.c_void_p()
    assert _lib.lib().olf_ctx_create(C.byref(p), 640, 480, 1, C.byref(h)) == _lib.OLF_ERR_INVALID
