"""GPU parity: ORBextractor on the MI355X vs the CPU oracle, bit-exact (key point coords, octave,
response, angle, 256-bit descriptors, order), through the C ABI."""
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth

pytestmark = pytest.mark.gpu

CONFIGS = [
    (640, 480, 1000),     # BASELINE config C2
    (1242, 375, 2000),    # C3 (KITTI)
    (752, 480, 1200),     # C4 (EuRoC)
]


def _compare(gk, gd, ok, od):
    assert len(gk) == len(ok), (len(gk), len(ok))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        a, b = gk[f], ok[f]
        assert np.array_equal(a.view(np.uint32) if a.dtype.kind == "f" else a, b.view(np.uint32) if b.dtype.kind == "f" else b), f
    assert np.array_equal(gd, od)


@pytest.mark.parametrize("w,h,nf", CONFIGS)
def test_orb_matches_oracle(oracle, w, h, nf):
    ex = ola.ORBextractor(nf, 1.2, 8, 20, 7)
    p = oracle.orb_params(nf)
    for seed in (1, 2):
        left, right = synth.stereo_pair(seed, w, h)
        for img in (left, right):
            gk, gd = ex(img)
            o = oracle.orb_extract(img, p, debug=True)
            # stage-by-stage so a mismatch names the first broken stage
            for l in range(8):
                assert np.array_equal(ex.pyramid_level(l), o["pyramid"][l]), f"pyramid level {l}"
            for l in range(8):
                assert np.array_equal(ex.debug_candidates(l), o["candidates"][l]), f"candidates level {l}"
            for l in range(8):
                if o["blurred"][l].any():
                    assert np.array_equal(ex.pyramid_level(l, blurred=True), o["blurred"][l]), f"blur level {l}"
            _compare(gk, gd, o["kps"], o["desc"])


def test_orb_batch_equals_single(oracle):
    w, h = 640, 480
    imgs = synth.stereo_batch(10, 3, w, h)
    ex = ola.ORBextractor(1000, 1.2, 8, 20, 7, max_images=6)
    kps, desc, counts = ex.extract_batch(imgs)
    p = oracle.orb_params(1000)
    for i in range(6):
        o = oracle.orb_extract(imgs[i], p)
        n = int(counts[i])
        _compare(kps[i, :n], desc[i, :n], o["kps"], o["desc"])


def test_orb_flat_image_has_no_keypoints():
    ex = ola.ORBextractor(500, 1.2, 8, 20, 7)
    k, d = ex(np.full((240, 320), 77, np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)


def test_orb_noise_and_1080p(oracle):
    """stress the octree (many candidates, equal-count ties) and the largest BASELINE size (1920x1080, 4000 features)"""
    rng = np.random.default_rng(1)
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    ex = ola.ORBextractor(1000, 1.2, 8, 20, 7)
    gk, gd = ex(noise)
    o = oracle.orb_extract(noise, oracle.orb_params(1000))
    _compare(gk, gd, o["kps"], o["desc"])
    left, _ = synth.stereo_pair(5, 1920, 1080)
    ex2 = ola.ORBextractor(4000, 1.2, 8, 20, 7)
    gk, gd = ex2(left)
    o = oracle.orb_extract(left, oracle.orb_params(4000), cap=5000)
    _compare(gk, gd, o["kps"], o["desc"])
    assert len(gk) >= 3900


@pytest.mark.parametrize("w,h,nf,sf,nl", [(1242, 375, 4500, 2.0, 2), (640, 480, 9000, 1.5, 3)])
def test_orb_levels_beyond_the_lds_octree(oracle, w, h, nf, sf, nl):
    """more than 2040 key points on one pyramid level (few levels, a large nfeatures): the octree lists spill to global memory; noise so that
    the quota is actually reached"""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[: h // 2] = synth.stereo_pair(3, w, h)[0][: h // 2]
    ex = ola.ORBextractor(nf, sf, nl, 20, 7)
    gk, gd = ex(img)
    o = oracle.orb_extract(img, oracle.orb_params(nf, sf, nl, 20, 7), cap=nf + 4096)
    _compare(gk, gd, o["kps"], o["desc"])
    assert (np.bincount(gk["octave"], minlength=nl) > 2040).any()
