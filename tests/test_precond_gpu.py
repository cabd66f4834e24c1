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
