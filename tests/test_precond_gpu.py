"""GPU parity: cvtColor(*2GRAY) and remap(INTER_LINEAR) (input conditioning) vs the CPU oracle."""
import numpy as np
import pytest
from orb_line_slam_amd import precond, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("code,cn", [(precond.RGB2GRAY, 3), (precond.BGR2GRAY, 3), (precond.RGBA2GRAY, 4), (precond.BGRA2GRAY, 4)])
def test_cvt_gray(oracle, code, cn):
    rng = np.random.default_rng(code)
    imgs = rng.integers(0, 256, (2, 375, 1242, cn), dtype=np.uint8)
    imgs[0, 0, :3] = [[255] * cn, [0] * cn, [1, 254, 7, 9][:cn]]
    g = precond.cvtColor(imgs, code)
    for i in range(2):
        assert np.array_equal(g[i], oracle.cvt_gray(imgs[i], code))
    assert g[0, 0, 0] == 255 and g[0, 0, 1] == 0


def test_remap_euroc_like(oracle):
    # EuRoC geometry: 752x480 input, rectifying maps of the same size built from a mild radial distortion + shift, incl. out-of-image taps
    w, h = 752, 480
    left, right = synth.stereo_pair(3, w, h)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    xn, yn = (xs - 367.2) / 458.6, (ys - 248.4) / 457.3
    r2 = xn * xn + yn * yn
    k = 1 + np.float32(-0.2834) * r2 + np.float32(0.0739) * r2 * r2
    mapx = (xn * k * np.float32(458.6) + np.float32(367.2) + np.float32(1.37)).astype(np.float32)
    mapy = (yn * k * np.float32(457.3) + np.float32(248.4) - np.float32(0.61)).astype(np.float32)
    mapx[:3, :5] = [-0.5, -1.0, -1.49, 751.4, 752.0]       # border cases: half-in, fully out, exact .5 roundings
    mapy[:3, :5] = [0.015625, 479.5, 480.2, -0.51, 100.984375]
    out = precond.remap(np.stack([left, right]), mapx, mapy)
    assert np.array_equal(out[0], oracle.remap_linear(left, mapx, mapy))
    assert np.array_equal(out[1], oracle.remap_linear(right, mapx, mapy))
    ident = precond.remap(left[None], xs, ys)
    assert np.array_equal(ident[0], left)                  # identity maps reproduce the image


def test_init_undistort_rectify_map_euroc(oracle):
    """cv::initUndistortRectifyMap with the EuRoC calibration of Examples/PL/PL_EuRoC.yaml (LEFT / RIGHT .K .D .R .P): both maps bit-identical
    to the oracle, and the rectified image through remap() equals the oracle's."""
    cams = {
        "left": (np.array([458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1.0]), np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0]),
                 np.array([0.999966347530033, -0.001422739138722922, 0.008079580483432283, 0.001365741834644127, 0.9999741760894847, 0.007055629199258132,
                           -0.008089410156878961, -0.007044357138835809, 0.9999424675829176]),
                 np.array([435.2046959714599, 0, 367.4517211914062, 0, 0, 435.2046959714599, 252.2008514404297, 0, 0, 0, 1, 0]).reshape(3, 4)),
        "right": (np.array([457.587, 0, 379.999, 0, 456.134, 255.238, 0, 0, 1.0]), np.array([-0.28368365, 0.07451284, -0.00010473, -3.555907e-05, 0.0]),
                  np.array([0.9999633526194376, -0.003625811871560086, 0.007755443660172947, 0.003680398547259526, 0.9999684752771629, -0.007035845251224894,
                            -0.007729688520722713, 0.007064130529506649, 0.999945173484644]),
                  np.array([435.2046959714599, 0, 367.4517211914062, -47.90639384423901, 0, 435.2046959714599, 252.2008514404297, 0, 0, 0, 1, 0]).reshape(3, 4)),
    }
    w, h = 752, 480
    left, right = synth.stereo_pair(3, w, h)
    for name, img in (("left", left), ("right", right)):
        K, D, R, P = cams[name]
        m1, m2 = precond.initUndistortRectifyMap(K, D, R, P, (w, h))
        o1, o2 = oracle.init_undistort_rectify_map(K, D, R, P, w, h)
        assert np.array_equal(m1.view(np.uint32), o1.view(np.uint32)) and np.array_equal(m2.view(np.uint32), o2.view(np.uint32)), name
        assert abs(float(m1[240, 376]) - 376) < 30 and abs(float(m2[240, 376]) - 240) < 30          # a rectification, not garbage
        assert np.array_equal(precond.remap(img[None], m1, m2)[0], oracle.remap_linear(img, o1, o2))
    # rational model (k4..k6) and no distortion at all
    K, _, R, P = cams["left"]
    for D in (np.array([0.1, -0.05, 0.001, -0.002, 0.01, 0.02, -0.01, 0.003]), np.zeros(0)):
        m1, m2 = precond.initUndistortRectifyMap(K, D, R, P, (320, 240))
        o1, o2 = oracle.init_undistort_rectify_map(K, D, R, P, 320, 240)
        assert np.array_equal(m1.view(np.uint32), o1.view(np.uint32)) and np.array_equal(m2.view(np.uint32), o2.view(np.uint32))


def test_colour_files_follow_camera_rgb(oracle, tmp_path):
    """Colour recordings as the reference's tracker sees them (src/Tracking.cc:193-218): cv::imread delivers B, G, R; Camera.RGB = 1 -- every KITTI / EuRoC yaml --
    applies CV_RGB2GRAY to that, i.e. 0.299 B + 0.587 G + 0.114 R; Camera.RGB = 0 applies CV_BGR2GRAY.  A synthetic colour PNG pair (and an RGBA one) in both
    settings against the oracle's cvtColor on the BGR(A)-ordered pixels; grey files pass unchanged."""
    from PIL import Image
    from orb_line_slam_amd import sequence
    rng = np.random.default_rng(3)
    h, w = 240, 320
    l, r = tmp_path / "left", tmp_path / "right"
    l.mkdir(); r.mkdir()
    rgb = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(2)]
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    Image.fromarray(rgb[0], "RGB").save(l / "000000.png"); Image.fromarray(rgb[1], "RGB").save(r / "000000.png")
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for camera_rgb, code in ((1, precond.RGB2GRAY), (0, precond.BGR2GRAY)):
        seq = sequence.StereoSequence(str(tmp_path), camera_rgb=camera_rgb)
        batch, _ = next(seq.batches(1))
        for side in (0, 1):
            bgr = np.ascontiguousarray(rgb[side][..., ::-1])
            assert np.array_equal(batch[side], oracle.cvt_gray(bgr, code)), (camera_rgb, side)
        # the two settings differ on colour input (the R and B weights swap)
    a = sequence.read_gray(str(l / "000000.png"), True); b = sequence.read_gray(str(l / "000000.png"), False)
    assert not np.array_equal(a, b)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "a.png")
    bgra = np.ascontiguousarray(np.concatenate([rgba[..., 2::-1], rgba[..., 3:]], axis=-1))
    assert np.array_equal(sequence.read_gray(str(tmp_path / "a.png"), True), oracle.cvt_gray(bgra, precond.RGBA2GRAY))
    assert np.array_equal(sequence.read_gray(str(tmp_path / "a.png"), False), oracle.cvt_gray(bgra, precond.BGRA2GRAY))
    Image.fromarray(grey, "L").save(tmp_path / "g.png")
    assert np.array_equal(sequence.read_gray(str(tmp_path / "g.png"), True), grey)
