"""GPU parity: Lineextractor (LSD + LBD), stereo line matching and the fused stereo-frame entry vs the CPU oracle."""
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth

pytestmark = pytest.mark.gpu
KL_FLOAT = ("angle", "pt_x", "pt_y", "response", "size", "startPointX", "startPointY", "endPointX", "endPointY", "sPointInOctaveX",
            "sPointInOctaveY", "ePointInOctaveX", "ePointInOctaveY", "lineLength")
KL_INT = ("class_id", "octave", "numOfPixels")


def _cmp_keylines(g, o):
    assert len(g) == len(o), (len(g), len(o))
    for f in KL_INT:
        assert np.array_equal(g[f], o[f]), f
    for f in KL_FLOAT:
        assert np.array_equal(g[f].view(np.uint32), o[f].view(np.uint32)), f


@pytest.mark.parametrize("w,h,nl", [(640, 480, 200), (1242, 375, 500), (752, 480, 0)])
def test_line_extract(oracle, w, h, nl):
    p = oracle.full_params(2000, nl)
    ex = ola.Lineextractor(nl, 0.025)
    for seed in (3, 4):
        left, right = synth.stereo_pair(seed, w, h)
        for img in (left, right):
            gk, gd = ex(img)
            _, scaled = oracle.lsd_detect(img, p.line)
            assert np.array_equal(ex.debug_scaled(), scaled), "LSD working image"
            o = oracle.line_extract(img, p.line)
            _cmp_keylines(gk, o["kls"])
            assert np.array_equal(gd, o["desc"])
            # convention C.3: the reference's unstable std::sort may permute equal responses; it must at least keep
            # the same multiset of responses (ties only), and agrees exactly when no response is repeated
            o2 = oracle.line_extract(img, p.line, use_std_sort=True)
            assert np.array_equal(np.sort(o2["kls"]["response"]), np.sort(o["kls"]["response"]))
            if len(np.unique(o["all"]["response"])) == len(o["all"]):
                assert np.array_equal(o2["kls"], o["kls"])


def test_lbd_on_given_keylines(oracle):
    w, h = 640, 480
    left, _ = synth.stereo_pair(9, w, h)
    p = oracle.full_params(1000, 200)
    o = oracle.line_extract(left, p.line)
    ex = ola.Lineextractor(200, 0.025)
    assert np.array_equal(ex.compute(left, o["kls"]), o["desc"])
    assert ex.compute(left, o["kls"][:0]).shape == (0, 32)


# BASELINE.json configs: C2 640x480 (1000 + 200), C3 KITTI 1242x375 (2000 + 500), C4 EuRoC 752x480, C5 1920x1080 (4000 + 1000)
@pytest.mark.parametrize("w,h,nf,nl,fx,bf,npairs", [(640, 480, 1000, 200, 435.2047, 47.9064, 3), (1242, 375, 2000, 500, 718.856, 386.1448, 3),
                                                    (752, 480, 1200, 500, 435.2047, 47.9064, 2), (1920, 1080, 4000, 1000, 1050.0, 126.0, 1)])       # C4: Examples/PL/PL_EuRoC.yaml:92,158 (1200 ORB, 500 lines)
def test_stereo_frames(oracle, w, h, nf, nl, fx, bf, npairs):
    p = oracle.full_params(nf, nl, fx, bf)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=npairs)
    imgs = synth.stereo_batch(31, npairs, w, h)
    f = fe.frames(imgs)
    for i in range(npairs):
        g = f.pair(i)
        o = oracle.stereo_points(imgs[2 * i], imgs[2 * i + 1], p)
        assert np.array_equal(g["mvKeys"], o["kpsL"]) and np.array_equal(g["mDescriptors"], o["descL"])
        assert np.array_equal(g["mvKeysRight"], o["kpsR"]) and np.array_equal(g["mDescriptorsRight"], o["descR"])
        assert np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32))
        assert np.array_equal(g["mvDepth"].view(np.uint32), o["depth"].view(np.uint32))
        ol, orr = oracle.line_extract(imgs[2 * i], p.line), oracle.line_extract(imgs[2 * i + 1], p.line)
        _cmp_keylines(g["mvKeys_Line"], ol["kls"])
        _cmp_keylines(g["mvKeysRight_Line"], orr["kls"])
        assert np.array_equal(g["mDescriptors_Line"], ol["desc"]) and np.array_equal(g["mDescriptorsRight_Line"], orr["desc"])
        m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
        assert np.array_equal(g["line_matches_12"], m)
        assert np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32))
        assert np.array_equal(g["mvle_l"].view(np.uint64), le.view(np.uint64))
        assert (m >= 0).sum() > 20 and (disp[:, 0] >= 0).sum() > 10


def test_stereo_frames_sweep_kitti(oracle):
    """wider seed sweep at the benchmark configuration (1242x375, 2000 ORB + 500 LBD): everything bit-identical for every pair"""
    w, h = 1242, 375
    p = oracle.full_params(2000, 500)
    n = 12
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=n)
    imgs = synth.stereo_batch(7000, n, w, h)          # the bench's own inputs
    f = fe.frames(imgs)
    for i in range(n):
        g = f.pair(i)
        o = oracle.stereo_points(imgs[2 * i], imgs[2 * i + 1], p)
        assert np.array_equal(g["mvKeys"], o["kpsL"]) and np.array_equal(g["mDescriptors"], o["descL"]), i
        assert np.array_equal(g["mvKeysRight"], o["kpsR"]) and np.array_equal(g["mDescriptorsRight"], o["descR"]), i
        assert np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32)), i
        assert np.array_equal(g["mvDepth"].view(np.uint32), o["depth"].view(np.uint32)), i
        ol, orr = oracle.line_extract(imgs[2 * i], p.line), oracle.line_extract(imgs[2 * i + 1], p.line)
        _cmp_keylines(g["mvKeys_Line"], ol["kls"])
        _cmp_keylines(g["mvKeysRight_Line"], orr["kls"])
        assert np.array_equal(g["mDescriptors_Line"], ol["desc"]) and np.array_equal(g["mDescriptorsRight_Line"], orr["desc"]), i
        m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
        assert np.array_equal(g["line_matches_12"], m), i
        assert np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32)), i
        assert np.array_equal(g["mvle_l"].view(np.uint64), le.view(np.uint64)), i


def test_line_edge_cases(oracle):
    ex = ola.Lineextractor(100, 0.025)
    k, d = ex(np.full((240, 320), 128, np.uint8))                     # flat image: no gradient, no lines, no crash
    assert len(k) == 0 and d.shape == (0, 32)
    img = np.zeros((240, 320), np.uint8); img[:, 160:] = 255          # one perfect vertical edge
    k, d = ex(img)
    p = oracle.full_params(500, 100)
    o = oracle.line_extract(img, p.line)
    _cmp_keylines(k, o["kls"])
    assert np.array_equal(d, o["desc"]) and len(k) >= 1
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (240, 320), dtype=np.uint8)          # pure noise: thousands of tiny regions
    k, d = ex(noise)
    o = oracle.line_extract(noise, p.line)
    _cmp_keylines(k, o["kls"])
    assert np.array_equal(d, o["desc"])


def _pattern(name, w, h):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    if name == "tri_x":        # triangle wave, slope 10 grey levels / pixel: 24-pixel wide bands of one gradient direction over the full height
        t = np.abs((x % 48) - 24) * 10
    elif name == "tri_diag":
        t = np.abs(((x + y) % 96) - 48) * 5
    elif name == "rings":      # curved level lines: the region angle drifts while a region grows
        r = np.hypot(x - w / 2, y - h / 2)
        t = np.abs((r % 64) - 32) * 7.5
    elif name == "checker":    # equal responses everywhere: ties in every selection
        t = (((x // 16) + (y // 16)) % 2) * 255
    elif name == "soft_edges": # wide smooth ramps between plateaus
        t = 127.5 + 127.5 * np.tanh((np.abs((x % 160) - 80) - 40) / 6.0)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("name", ["tri_x", "tri_diag", "rings", "checker", "soft_edges"])
def test_line_extract_structured_images(oracle, name):
    """regions of thousands of pixels (the growth FIFO leaves its LDS window and is read back from memory), drifting region angles,
    massive ties -- content the random synthetic scenes do not produce"""
    w, h = 640, 360
    img = _pattern(name, w, h)
    p = oracle.full_params(1000, 300)
    ex = ola.Lineextractor(300, 0.025)
    k, d = ex(img)
    o = oracle.line_extract(img, p.line)
    _cmp_keylines(k, o["kls"])
    assert np.array_equal(d, o["desc"])
    if name != "checker":
        assert len(k) > 0
    # and through the fused stereo entry (ORB side and both stereo matchers included), the pattern shifted by 5 px as the right image
    pq = oracle.full_params(1000, 300, 400.0, 40.0)
    fe = ola.StereoFrontEnd(pq, w, h, max_pairs=1)
    pair = np.stack([img, np.roll(img, -5, axis=1)])
    g = fe.frames(pair).pair(0)
    op = oracle.stereo_points(pair[0], pair[1], pq)
    assert np.array_equal(g["mvKeys"], op["kpsL"]) and np.array_equal(g["mDescriptors"], op["descL"])
    assert np.array_equal(g["mvKeysRight"], op["kpsR"]) and np.array_equal(g["mDescriptorsRight"], op["descR"])
    assert np.array_equal(g["mvuRight"].view(np.uint32), op["uRight"].view(np.uint32))
    ol, orr = oracle.line_extract(pair[0], pq.line), oracle.line_extract(pair[1], pq.line)
    _cmp_keylines(g["mvKeys_Line"], ol["kls"])
    _cmp_keylines(g["mvKeysRight_Line"], orr["kls"])
    m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, pq.stereo)
    assert np.array_equal(g["line_matches_12"], m) and np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32))


def test_line_extract_wide_growth_front(oracle):
    """a shallow ramp under a low gradient threshold, LSD working at twice the input size: one direction over 170-pixel wide bands, so the growth front holds several hundred
    pending pixels and the FIFO window leaves the agent's 256-entry LDS ring (entries are then read back from the region log in memory)"""
    w, h = 680, 300
    y, x = np.mgrid[0:h, 0:w]
    img = np.clip(np.abs((x % 170) - 85) * 3, 0, 255).astype(np.uint8)
    p = oracle.full_params(500, 100)
    p.line.lsd_quant, p.line.lsd_scale = 0.3, 2.0
    ex = ola.Lineextractor(100, 0.025, lsd_quant=0.3, lsd_scale=2.0)
    k, d = ex(img)
    o = oracle.line_extract(img, p.line)
    _cmp_keylines(k, o["kls"])
    assert np.array_equal(d, o["desc"]) and len(k) > 0


@pytest.mark.parametrize("scale", [0.65, 0.575])
def test_lsd_wide_blur_saturated_regions(oracle, scale):
    """lsd_scale 0.65 / 0.575 at sigma_scale 0.6: the 9-tap kernels' independently rounded 8-bit taps sum to 259 / 258, so the row sums of
    a 255-valued area (259 * 255 = 66 045) do not fit 16 bits -- the wide blur keeps them in 32 bits like the oracle"""
    w, h = 640, 360
    y, x = np.mgrid[0:h, 0:w]
    img = np.where(((x // 80) + (y // 60)) % 2 == 0, 255, 40).astype(np.uint8)      # saturated plateaus with long straight edges
    img[100:140, 200:420] = 255
    p = oracle.full_params(500, 100)
    p.line.lsd_scale = scale
    ex = ola.Lineextractor(100, 0.025, lsd_scale=scale)
    k, d = ex(img)
    _, scaled = oracle.lsd_detect(img, p.line)
    assert np.array_equal(ex.debug_scaled(), scaled), "LSD working image"
    assert scaled.max() >= 250                                                        # bright areas stay bright
    o = oracle.line_extract(img, p.line)
    _cmp_keylines(k, o["kls"])
    assert np.array_equal(d, o["desc"]) and len(k) > 0


@pytest.mark.parametrize("w,h", [(333, 257), (1000, 300), (401, 243), (897, 601)])
def test_stereo_frames_odd_sizes(oracle, w, h):
    """sizes that are multiples of nothing: tile edges, pitch padding, level geometry, cell grids with one column, tiny top levels"""
    p = oracle.full_params(700, 150, 400.0, 40.0)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=2)
    imgs = synth.stereo_batch(w + h, 2, w, h)
    f = fe.frames(imgs)
    for i in range(2):
        g = f.pair(i)
        o = oracle.stereo_points(imgs[2 * i], imgs[2 * i + 1], p)
        assert np.array_equal(g["mvKeys"], o["kpsL"]) and np.array_equal(g["mDescriptors"], o["descL"])
        assert np.array_equal(g["mvKeysRight"], o["kpsR"]) and np.array_equal(g["mDescriptorsRight"], o["descR"])
        assert np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32))
        ol, orr = oracle.line_extract(imgs[2 * i], p.line), oracle.line_extract(imgs[2 * i + 1], p.line)
        _cmp_keylines(g["mvKeys_Line"], ol["kls"])
        _cmp_keylines(g["mvKeysRight_Line"], orr["kls"])
        assert np.array_equal(g["mDescriptors_Line"], ol["desc"]) and np.array_equal(g["mDescriptorsRight_Line"], orr["desc"])
        m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
        assert np.array_equal(g["line_matches_12"], m) and np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32))


@pytest.mark.parametrize("orb,line", [
    ((1500, 1.5, 4, 25, 10), dict(lsd_nfeatures=0, min_line_length=0.05, lsd_ang_th=20.0, lsd_n_bins=512)),      # keep all lines
    ((300, 1.1, 6, 12, 5), dict(lsd_nfeatures=80, lsd_scale=1.0, lsd_quant=1.5)),                                  # no LSD rescale
    ((800, 1.2, 8, 30, 15), dict(lsd_nfeatures=150, lsd_scale=0.8, lsd_sigma_scale=0.6, lsd_density_th=0.7)),       # OpenCV's default LSD scale
    ((600, 1.2, 8, 20, 7), dict(lsd_nfeatures=100, lsd_scale=0.8, lsd_sigma_scale=0.75)),                           # 9-tap LSD blur (k_sep_wide)
    ((600, 1.2, 8, 20, 7), dict(lsd_nfeatures=0, lsd_scale=0.5, lsd_sigma_scale=0.75, min_line_length=0.02)),       # 13-tap blur, half-size working image
])
def test_stereo_frames_parameter_sets(oracle, orb, line):
    """non-default ORBextractor / Config parameters: other pyramid geometry, thresholds, LSD scale (incl. down-scaling) and bins"""
    w, h = 640, 480
    p = oracle.full_params(orb[0], 0, 500.0, 60.0)
    p.orb.nfeatures, p.orb.scale_factor, p.orb.nlevels, p.orb.ini_th_fast, p.orb.min_th_fast = orb
    for k, v in line.items():
        setattr(p.line, k, v)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=2)
    imgs = synth.stereo_batch(77 + orb[0], 2, w, h)
    f = fe.frames(imgs)
    for i in range(2):
        g = f.pair(i)
        o = oracle.stereo_points(imgs[2 * i], imgs[2 * i + 1], p)
        assert np.array_equal(g["mvKeys"], o["kpsL"]) and np.array_equal(g["mDescriptors"], o["descL"])
        assert np.array_equal(g["mvKeysRight"], o["kpsR"]) and np.array_equal(g["mDescriptorsRight"], o["descR"])
        assert np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32))
        ol, orr = oracle.line_extract(imgs[2 * i], p.line), oracle.line_extract(imgs[2 * i + 1], p.line)
        _cmp_keylines(g["mvKeys_Line"], ol["kls"])
        _cmp_keylines(g["mvKeysRight_Line"], orr["kls"])
        assert np.array_equal(g["mDescriptors_Line"], ol["desc"]) and np.array_equal(g["mDescriptorsRight_Line"], orr["desc"])
        m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
        assert np.array_equal(g["line_matches_12"], m) and np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32))
        assert len(ol["kls"]) > (20 if line.get("lsd_scale", 1.2) >= 0.8 else 5)


def test_context_reuse_across_batches(oracle):
    """one context, batches of different sizes and content back to back: no state leaks from one call into the next"""
    w, h = 640, 480
    p = oracle.full_params(1000, 200, 435.2047, 47.9064)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=3)
    fresh = lambda imgs: ola.StereoFrontEnd(p, w, h, max_pairs=len(imgs) // 2).frames(imgs)
    flat = np.full((2, h, w), 90, np.uint8)
    for seed, npairs in [(5, 3), (6, 1), (0, 1), (7, 2), (5, 3)]:
        imgs = flat if seed == 0 else synth.stereo_batch(seed, npairs, w, h)
        a, b = fe.frames(imgs), fresh(imgs)
        for i in range(npairs):
            ga, gb = a.pair(i), b.pair(i)
            for k in ga:
                assert ga[k].tobytes() == gb[k].tobytes(), (seed, i, k)
    o = oracle.stereo_points(imgs[0], imgs[1], p)
    assert np.array_equal(a.pair(0)["mvKeys"], o["kpsL"])


def test_offline_pipeline_equals_frames(oracle):
    """the double-buffered host pipeline (H2D / path / D2H on three streams) returns exactly what StereoFrontEnd.frames returns"""
    from orb_line_slam_amd.pipeline import OfflinePipeline
    w, h = 640, 480
    p = oracle.full_params(1000, 200, 435.2047, 47.9064)
    pipe = OfflinePipeline(p, w, h, pairs_per_batch=3)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=3)
    batches = [synth.stereo_batch(100 + i, 3 if i != 3 else 1, w, h) for i in range(6)]
    got = 0
    for i, f in enumerate(pipe.run(batches)):
        ref = fe.frames(batches[i])
        for j in range(batches[i].shape[0] // 2):
            a, b = f.pair(j), ref.pair(j)
            for k in a:
                assert np.ascontiguousarray(a[k]).tobytes() == np.ascontiguousarray(b[k]).tobytes(), (i, j, k)
        got += 1
    assert got == len(batches)


def test_stereo_frame_against_committed_fixture():
    """The HIP path against tests/golden/frame_320x240_seed11.npz directly (no oracle library in the loop): the committed outputs of the
    whole feature path for one seeded stereo pair."""
    import os, zlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frame_320x240_seed11.npz"))
    left, right = synth.stereo_pair(11, 320, 240)
    assert zlib.crc32(left.tobytes()) == int(g["crc_left"]) and zlib.crc32(right.tobytes()) == int(g["crc_right"])
    from orb_line_slam_amd import _lib
    p = _lib.default_params()
    p.orb.nfeatures, p.line.lsd_nfeatures, p.stereo.fx, p.stereo.bf = 500, 100, 300.0, 40.0
    fe = ola.StereoFrontEnd(p, 320, 240, max_pairs=1)
    f = fe.frames(np.stack([left, right])).pair(0)
    assert np.array_equal(f["mvKeys"], g["kpsL"]) and np.array_equal(f["mDescriptors"], g["descL"])
    assert np.array_equal(f["mvuRight"].view(np.uint32), g["uRight"].view(np.uint32)) and np.array_equal(f["mvDepth"].view(np.uint32), g["depth"].view(np.uint32))
    assert np.array_equal(f["mvKeys_Line"], g["klsL"]) and np.array_equal(f["mDescriptors_Line"], g["ldescL"])
    assert np.array_equal(f["line_matches_12"], g["lm12"]) and np.array_equal(f["mvDisparity_l"].view(np.uint32), g["ldisp"].view(np.uint32))
    assert np.array_equal(f["mvle_l"].view(np.uint64), g["lle"].view(np.uint64))


@pytest.mark.parametrize("gauss256,exact", [(1, 0), (0, 1), (1, 1)])
def test_opencv_version_conventions(oracle, gauss256, exact):
    """Conventions C.10 / C.11 (which OpenCV 3.4.x patch level the reference linked decides them): the library follows the oracle under every
    combination, and under the raster seed order of OpenCV <= 3.2 (C.9 variant 0; variant 1, std::sort, is the default and what every other test runs)."""
    from orb_line_slam_amd import _lib
    w, h = 640, 480
    p = oracle.full_params(1000, 200, 435.2047, 47.9064)
    p.orb.conv_gauss_sum256 = p.line.conv_gauss_sum256 = gauss256
    p.line.conv_resize_exact = exact
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=1)
    imgs = synth.stereo_batch(77, 1, w, h)
    g = fe.frames(imgs).pair(0)
    o = oracle.stereo_points(imgs[0], imgs[1], p)
    assert np.array_equal(g["mvKeys"], o["kpsL"]) and np.array_equal(g["mDescriptors"], o["descL"])
    assert np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32))
    ol = oracle.line_extract(imgs[0], p.line)
    _cmp_keylines(g["mvKeys_Line"], ol["kls"])
    assert np.array_equal(g["mDescriptors_Line"], ol["desc"])
    base = oracle.full_params(1000, 200, 435.2047, 47.9064)
    assert not np.array_equal(ol["desc"], oracle.line_extract(imgs[0], base.line)["desc"]) or not np.array_equal(o["descL"], oracle.stereo_points(imgs[0], imgs[1], base)["descL"])
    # the raster seed order (OpenCV <= 3.2) through the fused entry, against the oracle under the same convention
    p.line.conv_seed_order = 0
    g0 = ola.StereoFrontEnd(p, w, h, max_pairs=1).frames(imgs).pair(0)
    ol0 = oracle.line_extract(imgs[0], p.line)
    _cmp_keylines(g0["mvKeys_Line"], ol0["kls"])
    assert np.array_equal(g0["mDescriptors_Line"], ol0["desc"])
    p.line.conv_seed_order = 2
    with pytest.raises(_lib.OlfError):
        ola.StereoFrontEnd(p, w, h, max_pairs=1)


def test_line_extract_long_line_scene(oracle):
    """the 'bars' synthetic scene (key lines of about 0.08 * W pixels, the length SURVEY App. D models for a KITTI frame): LBD support regions
    three times longer than in the default scene, region2rect over regions of several hundred pixels"""
    w, h = 1242, 375
    p = oracle.full_params(2000, 500)
    ex = ola.Lineextractor(500, 0.025)
    left, right = synth.stereo_pair(5, w, h, scene="bars")
    for img in (left, right):
        gk, gd = ex(img)
        o = oracle.line_extract(img, p.line)
        _cmp_keylines(gk, o["kls"])
        assert np.array_equal(gd, o["desc"])
        assert gk["numOfPixels"].mean() > 90


def test_line_extract_full_occupancy_batch(oracle):
    """6144 KITTI-sized images in one call -- the bench's batch: 24 one-wave seed sorts / growth agents per CU, i.e. workgroups whose LDS
    allocations start far above 64 KB (the seed sort stages blocks through global_load_lds, whose M0 base has 16 bits), every resident wave
    slot taken.  Every image's key lines and descriptors must equal the oracle's for the distinct image it is a copy of."""
    w, h, distinct, n = 1242, 375, 24, 6144
    base = synth.stereo_batch(7100, distinct // 2, w, h)
    p = oracle.full_params(2000, 500)
    want = [oracle.line_extract(im, p.line) for im in base]
    imgs = np.tile(base, (n // distinct, 1, 1))
    assert len(imgs) == n
    ex = ola.Lineextractor(500, 0.025, max_images=n)
    kls, desc, counts = ex.extract_batch(imgs)
    bad = []
    for i in range(n):
        o = want[i % distinct]
        c = int(counts[i])
        if c != len(o["kls"]) or not np.array_equal(kls[i, :c], o["kls"]) or not np.array_equal(desc[i, :c], o["desc"]):
            bad.append(i)
    ex._ctx.close()      # (a context of this size holds > 100 GB: released here, not whenever the collector gets to it)
    assert not bad, (len(bad), bad[:8])


def test_stereo_frames_full_batch(oracle):
    """the headline batch -- 3072 pairs through the fused entry (two streams, the ORB kernels in the growth agents' shadow, every wave slot and
    most of the LDS taken): every pair's key points, descriptors, stereo matches, key lines and line matches equal the oracle's for the
    distinct pair it is a copy of"""
    w, h, distinct, n = 1242, 375, 8, 3072
    p = oracle.full_params(2000, 500, 718.856, 386.1448)
    base = synth.stereo_batch(7300, distinct, w, h)
    want = []
    for i in range(distinct):
        o = oracle.stereo_points(base[2 * i], base[2 * i + 1], p)
        ol, orr = oracle.line_extract(base[2 * i], p.line), oracle.line_extract(base[2 * i + 1], p.line)
        m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
        want.append((o, ol, orr, m, disp, le))
    imgs = np.tile(base, (n // distinct, 1, 1))
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=n)
    f = fe.frames(imgs)
    bad = []
    for i in range(n):
        g = f.pair(i)
        o, ol, orr, m, disp, le = want[i % distinct]
        ok = (np.array_equal(g["mvKeys"], o["kpsL"]) and np.array_equal(g["mDescriptors"], o["descL"]) and np.array_equal(g["mvKeysRight"], o["kpsR"])
              and np.array_equal(g["mDescriptorsRight"], o["descR"]) and np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32))
              and np.array_equal(g["mvDepth"].view(np.uint32), o["depth"].view(np.uint32))
              and np.array_equal(g["mvKeys_Line"], ol["kls"]) and np.array_equal(g["mvKeysRight_Line"], orr["kls"])
              and np.array_equal(g["mDescriptors_Line"], ol["desc"]) and np.array_equal(g["mDescriptorsRight_Line"], orr["desc"])
              and np.array_equal(g["line_matches_12"], m) and np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32))
              and np.array_equal(g["mvle_l"].view(np.uint64), le.view(np.uint64)))
        if not ok:
            bad.append(i)
    fe.ctx.close()
    assert not bad, (len(bad), bad[:8])


@pytest.mark.parametrize("libm_float,eigen_recip", [(1, 0), (0, 1), (1, 1)])
def test_convention_variants_libm_float_and_eigen_reciprocal(oracle, libm_float, eigen_recip):
    """Conventions C.6 / Eigen (round 4): conv_libm_float = 1 -- cosf / sinf in region_grow and LBD, atan2f for KeyLine.angle, float 1 / sqrtf in the LBD
    normalisation -- and conv_eigen_recip = 1 -- mvle_l = le_l * (1 / norm) -- each restated in the oracle and on the device: the fused stereo entry must
    equal the oracle bit for bit under every combination, and the variants must really differ from the defaults somewhere (or the switch is dead)."""
    w, h = 640, 480
    p = oracle.full_params(1000, 200, 435.2047, 47.9064)
    p.line.conv_libm_float, p.stereo.conv_eigen_recip = libm_float, eigen_recip
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=2)
    imgs = synth.stereo_batch(31, 2, w, h)
    fr = fe.frames(imgs)
    p0 = oracle.full_params(1000, 200, 435.2047, 47.9064)
    differs = False
    for q in range(2):
        g = fr.pair(q)
        ol, orr = oracle.line_extract(imgs[2 * q], p.line), oracle.line_extract(imgs[2 * q + 1], p.line)
        assert np.array_equal(g["mvKeys_Line"], ol["kls"]) and np.array_equal(g["mDescriptors_Line"], ol["desc"]), (q, "left lines")
        assert np.array_equal(g["mvKeysRight_Line"], orr["kls"]) and np.array_equal(g["mDescriptorsRight_Line"], orr["desc"]), (q, "right lines")
        m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
        assert np.array_equal(g["line_matches_12"], m) and np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32))
        assert np.array_equal(g["mvle_l"].view(np.uint64), le.view(np.uint64)), (q, "mvle_l")
        o0 = oracle.line_extract(imgs[2 * q], p0.line)
        o0r = oracle.line_extract(imgs[2 * q + 1], p0.line)
        if libm_float and (len(o0["kls"]) != len(ol["kls"]) or not np.array_equal(o0["kls"], ol["kls"]) or not np.array_equal(o0["desc"], ol["desc"])):
            differs = True
        if eigen_recip and not libm_float:
            _, _, le0 = oracle.stereo_lines(o0["kls"], o0["desc"], o0r["kls"], o0r["desc"], w, h, p0.stereo)
            differs = differs or not np.array_equal(le0.view(np.uint64), le.view(np.uint64))
    assert differs, "the convention switch changed nothing on these images"


def test_pipelined_batches_with_input_event(oracle):
    """olf_ctx_set_input_event: with the caller's input-ready event the line stream of a call no longer forks from the caller's stream, so the LSD front
    of batch k + 1 runs beside the ORB / stereo tail of batch k (and beside whatever the caller queued behind it -- here device-side copies of every
    output).  Three batches back to back on one context and ONE set of output buffers, alternating between two resident inputs, must give what the
    same calls give one at a time."""
    import ctypes as C
    import torch
    from orb_line_slam_amd import _lib
    from orb_line_slam_amd._lib import FrameBuffers, check as chk, lib
    w, h, n = 1242, 375, 192
    p = oracle.full_params(2000, 500, 718.856, 386.1448)
    dev = torch.device("cuda", 0)
    inputs = [torch.from_numpy(np.tile(synth.stereo_batch(7400 + 10 * k, 8, w, h), (n // 8, 1, 1))).to(dev) for k in range(2)]
    ctx = _lib.Context(p, w, h, 2 * n)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity
    spec = [((2 * n, cap, 28), torch.uint8), ((2 * n, cap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, cap), torch.float32), ((n, cap), torch.float32),
            ((2 * n, lcap, 68), torch.uint8), ((2 * n, lcap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, lcap), torch.int32), ((n, lcap, 2), torch.float32),
            ((n, lcap, 3), torch.float64)]
    out = [torch.zeros(sh, dtype=dt, device=dev) for sh, dt in spec]
    fb = FrameBuffers(*[t.data_ptr() for t in out])
    # (a stream of the caller's own: torch's default stream has the handle NULL, which the C ABI reads as "the context's stream" -- the copies below
    # would then run on a stream that is not ordered with the calls)
    st = torch.cuda.Stream(dev)
    s = st.cuda_stream
    assert s != 0
    order = [0, 1, 0, 1]

    def run(pipelined):
        got = []
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for t in out:
                t.zero_()                                     # (rows past a count keep what the previous call left there: both runs start from the same state)
            ev = torch.cuda.Event(); ev.record(st); torch.cuda.synchronize()
            ctx.set_input_event(ev if pipelined else None)
            for k in order:
                chk(lib().olf_stereo_frames_dev(ctx.handle, inputs[k].data_ptr(), n, C.byref(fb), s), "olf_stereo_frames_dev")
                got.append([t.clone() for t in out])          # on the caller's stream, behind the call: the next call's line stream must not overtake it
                if not pipelined:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
        ctx.synchronize()
        ctx.set_input_event(None)
        return [[t.cpu().numpy() for t in g] for g in got]

    ref, pip = run(False), run(True)
    for i in range(len(order)):
        for a, b in zip(ref[i], pip[i]):
            assert a.tobytes() == b.tobytes(), i
    assert ref[0][2].sum() > 1000 * n and not np.array_equal(ref[0][0], ref[1][0])      # key points were found (by the FIRST call already), and the two inputs differ
    assert np.array_equal(ref[0][2], ref[2][2]) and np.array_equal(ref[0][7], ref[2][7])      # (counts; rows past a count keep the previous batch's bytes)
    assert np.array_equal(ref[1][2], ref[3][2]) and not np.array_equal(ref[0][2], ref[1][2])
    ctx.close()


def test_deferred_join(oracle):
    """olf_ctx_set_deferred_join: olf_stereo_frames_dev returns with the point-feature outputs complete on the caller's stream and the line path's tail still running
    on the context's line stream; olf_stereo_frames_join_dev makes the stream wait for it.  Copies of the ORB outputs taken on the stream BEFORE the join and of the line
    outputs taken behind it must equal what a joined call leaves; a second call without an explicit join in between joins by itself."""
    import ctypes as C
    import torch
    from orb_line_slam_amd import _lib
    from orb_line_slam_amd._lib import FrameBuffers, check as chk, lib
    w, h, n = 1242, 375, 96
    p = oracle.full_params(2000, 500, 718.856, 386.1448)
    dev = torch.device("cuda", 0)
    inputs = [torch.from_numpy(np.tile(synth.stereo_batch(7500 + 10 * k, 8, w, h), (n // 8, 1, 1))).to(dev) for k in range(2)]
    ctx = _lib.Context(p, w, h, 2 * n)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity
    spec = [((2 * n, cap, 28), torch.uint8), ((2 * n, cap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, cap), torch.float32), ((n, cap), torch.float32),
            ((2 * n, lcap, 68), torch.uint8), ((2 * n, lcap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, lcap), torch.int32), ((n, lcap, 2), torch.float32),
            ((n, lcap, 3), torch.float64)]
    out = [torch.zeros(sh, dtype=dt, device=dev) for sh, dt in spec]
    fb = FrameBuffers(*[t.data_ptr() for t in out])
    st = torch.cuda.Stream(dev)

    def run(deferred):
        got = []
        torch.cuda.synchronize()
        chk(lib().olf_ctx_set_deferred_join(ctx.handle, 1 if deferred else 0), "olf_ctx_set_deferred_join")
        with torch.cuda.stream(st):
            for t in out:
                t.zero_()
            for k in (0, 1):
                chk(lib().olf_stereo_frames_dev(ctx.handle, inputs[k].data_ptr(), n, C.byref(fb), st.cuda_stream), "olf_stereo_frames_dev")
                orb = [t.clone() for t in out[:5]]                # before the join: the point features are complete on the stream
                if k == 0:
                    chk(lib().olf_stereo_frames_join_dev(ctx.handle, st.cuda_stream), "olf_stereo_frames_join_dev")
                    got.append(orb + [t.clone() for t in out[5:]])
                else:
                    got.append(orb)                               # (no join: the next call, or the one below, does it)
            chk(lib().olf_stereo_frames_join_dev(ctx.handle, st.cuda_stream), "olf_stereo_frames_join_dev")
            got[1] = got[1] + [t.clone() for t in out[5:]]
            torch.cuda.synchronize()
        ctx.synchronize()
        return [[t.cpu().numpy() for t in g] for g in got]

    ref, dfd = run(False), run(True)
    for i in range(2):
        for a, b in zip(ref[i], dfd[i]):
            assert a.tobytes() == b.tobytes(), i
    assert ref[0][2].sum() > 1000 * n and ref[0][7].sum() > 100 * n
    chk(lib().olf_ctx_set_deferred_join(ctx.handle, 0), "olf_ctx_set_deferred_join")
    ctx.close()


def test_stale_input_event_and_unjoined_readers(oracle):
    """ADVICE r4.  (a) olf_ctx_set_input_event is one-shot and the host entry olf_stereo_frames ignores it: with an event left behind by an earlier caller the line
    stream used to wait for that (long signalled) event only and could read c->d_images before the upload queued on the context's stream had landed.  (b) With the
    deferred join on, olf_frames_pack_dev and olf_match_bf_dev on the line descriptors join by themselves instead of reading what the line stream is still writing."""
    import ctypes as C
    import torch
    from orb_line_slam_amd import _lib
    from orb_line_slam_amd._lib import FrameBuffers, check as chk, lib
    w, h, n = 1242, 375, 48
    p = oracle.full_params(2000, 500, 718.856, 386.1448)
    dev = torch.device("cuda", 0)
    host = [np.tile(synth.stereo_batch(9100 + 10 * k, 4, w, h), (n // 4, 1, 1)) for k in range(2)]
    # (a) the host entry with a stale event set: results of alternating inputs must equal those of a clean context
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=n)
    clean = [fe.frames(host[k]) for k in (0, 1)]
    ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
    for k in (1, 0, 1):
        fe.ctx.set_input_event(ev)                       # signalled long ago; the caller then forgets about it
        f = fe.frames(host[k])
        assert np.array_equal(f.N_l, clean[k].N_l) and f.mvKeys_Line.tobytes() == clean[k].mvKeys_Line.tobytes(), k
        assert f.mDescriptors_Line.tobytes() == clean[k].mDescriptors_Line.tobytes() and f.line_matches_12.tobytes() == clean[k].line_matches_12.tobytes(), k
    # (b) deferred join, then the packer / the line matcher without olf_stereo_frames_join_dev
    ctx = _lib.Context(p, w, h, 2 * n)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity
    spec = [((2 * n, cap, 28), torch.uint8), ((2 * n, cap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, cap), torch.float32), ((n, cap), torch.float32),
            ((2 * n, lcap, 68), torch.uint8), ((2 * n, lcap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, lcap), torch.int32), ((n, lcap, 2), torch.float32),
            ((n, lcap, 3), torch.float64)]
    out = [torch.zeros(sh, dtype=dt, device=dev) for sh, dt in spec]
    fb = FrameBuffers(*[t.data_ptr() for t in out])
    bound = lib().olf_frames_pack_bound(ctx.handle, n)
    packed = torch.zeros(bound, dtype=torch.uint8, device=dev); nbytes = torch.zeros(1, dtype=torch.int64, device=dev)
    m12 = torch.zeros((n - 1, 2 * lcap), dtype=torch.int32, device=dev)
    imgs = torch.from_numpy(host[0]).to(dev)
    st = torch.cuda.Stream(dev)
    res = {}
    for deferred in (0, 1):
        torch.cuda.synchronize()
        chk(lib().olf_ctx_set_deferred_join(ctx.handle, deferred), "olf_ctx_set_deferred_join")
        with torch.cuda.stream(st):
            for t in out + [packed, m12]:
                t.zero_()
            chk(lib().olf_stereo_frames_dev(ctx.handle, imgs.data_ptr(), n, C.byref(fb), st.cuda_stream), "olf_stereo_frames_dev")
            chk(lib().olf_match_bf_dev(ctx.handle, out[6].data_ptr() + 2 * lcap * 32, out[7].data_ptr() + 8, 2 * lcap, 2, out[6].data_ptr(), out[7].data_ptr(),
                                       2 * lcap, 2, n - 1, 0.75, 1, m12.data_ptr(), st.cuda_stream), "olf_match_bf_dev")
            a = m12.clone()
        torch.cuda.synchronize(); ctx.synchronize()
        with torch.cuda.stream(st):
            chk(lib().olf_stereo_frames_dev(ctx.handle, imgs.data_ptr(), n, C.byref(fb), st.cuda_stream), "olf_stereo_frames_dev")
            chk(lib().olf_frames_pack_dev(ctx.handle, C.byref(fb), n, packed.data_ptr(), bound, nbytes.data_ptr(), st.cuda_stream), "olf_frames_pack_dev")
            b = packed.clone()
        torch.cuda.synchronize(); ctx.synchronize()
        res[deferred] = (a.cpu().numpy().tobytes(), b.cpu().numpy().tobytes(), int(nbytes.item()))
    assert res[0] == res[1] and res[0][2] > 0
    chk(lib().olf_ctx_set_deferred_join(ctx.handle, 0), "olf_ctx_set_deferred_join")
    ctx.close()
