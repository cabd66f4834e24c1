"""GPU parity of the multi-wave LSD region growing (csrc/lsd_grow.hip): the detected key lines must not depend on how many waves grow an
image concurrently nor on the size of the reorder buffer, and must equal the oracle's sequential seed loop (oracle/line_oracle.cpp:121-148)."""
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib

pytestmark = pytest.mark.gpu


def _set(ex, w, h, n, waves, rob):
    ctx = ex._context(w, h, n)
    _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, waves, rob), "olf_debug_lsd_waves")


def _status(ex):
    out = np.zeros(64, np.int32)
    _lib.check(_lib.lib().olf_debug_status(ex._ctx.handle, _lib.ptr(out)), "olf_debug_status")
    return out


@pytest.mark.parametrize("w,h", [(640, 480), (1242, 375)])
def test_growth_is_independent_of_wave_count(oracle, w, h):
    p = oracle.full_params(2000, 0)
    imgs = synth.stereo_batch(71, 2, w, h)            # 4 images
    want = [oracle.line_extract(im, p.line) for im in imgs]
    ex = ola.Lineextractor(0, 0.025, max_images=4)
    for waves, rob in [(0, 0), (1, 128), (2, 128), (2, 256), (4, 256), (8, 512), (16, 512), (16, 128), (3, 256), (-1, 0)]:
        _set(ex, w, h, 4, waves, rob)
        kls, desc, counts = ex.extract_batch(imgs)
        assert (_status(ex)[0] & (8 | 16)) == 0, (waves, rob, "capacity / watchdog flag")
        for i in range(4):
            n = int(counts[i])
            assert n == len(want[i]["kls"]), (waves, rob, i, n, len(want[i]["kls"]))
            assert np.array_equal(kls[i, :n], want[i]["kls"]), (waves, rob, i)
            assert np.array_equal(desc[i, :n], want[i]["desc"]), (waves, rob, i)


def test_growth_noise_and_flat_images(oracle):
    """Pure noise (hundreds of thousands of tiny regions, the chunk pool's worst case that still fits) and a flat image (no seed at all)."""
    w, h = 320, 240
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    flat = np.full((h, w), 77, np.uint8)
    p = oracle.full_params(1000, 0)
    ex = ola.Lineextractor(0, 0.025, max_images=2)
    for waves in (16, 4, 1):
        _set(ex, w, h, 2, waves, 0)
        for img in (noise, flat):
            gk, gd = ex(img)
            o = oracle.line_extract(img, p.line)
            assert np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), waves


def test_pool_exhaustion_falls_back_to_the_one_wave_agent(oracle):
    """an image that exhausts the multi-wave kernel's chunk pool used to lose all its lines (OLF_ERR_CAPACITY).  With the pool capped
    far below what the image needs, every image must come out identical to the oracle -- grown again by the one-wave agent in the same call --
    and no capacity flag may be raised; a mixed batch (one image that fits the capped pool, three that do not) exercises k_lsd_rect_mixed."""
    w, h = 640, 480
    p = oracle.full_params(2000, 0)
    imgs = synth.stereo_batch(71, 2, w, h)
    imgs[3] = 90                                           # a flat image: no regions at all, stays on the chunk-chain path
    want = [oracle.line_extract(im, p.line) for im in imgs]
    ex = ola.Lineextractor(0, 0.025, max_images=4)
    for waves, rob, pool in [(8, 256, 300), (16, 512, 600), (4, 128, 2000), (8, 256, 0)]:
        _set(ex, w, h, 4, waves, rob)
        _lib.check(_lib.lib().olf_debug_lsd_pool(ex._ctx.handle, pool), "olf_debug_lsd_pool")
        kls, desc, counts = ex.extract_batch(imgs)
        assert (_status(ex)[0] & (8 | 16)) == 0, (waves, rob, pool)
        for i in range(4):
            n = int(counts[i])
            assert n == len(want[i]["kls"]), (waves, pool, i, n, len(want[i]["kls"]))
            assert np.array_equal(kls[i, :n], want[i]["kls"]) and np.array_equal(desc[i, :n], want[i]["desc"]), (waves, pool, i)
    _lib.check(_lib.lib().olf_debug_lsd_pool(ex._ctx.handle, 0), "olf_debug_lsd_pool")


def test_pool_fallback_keeps_one_stride_for_both_pixel_list_formats(oracle):
    """the chunk chains of the multi-wave kernel and the contiguous log of the one-wave fall-back used different per-image strides in the
    shared pixel-list buffer unless Ps % 16 == 0 and Ps >= 34816.  A small working image (180 x 133, Ps = 23940, Ps % 16 = 4), chained images WITH
    regions before and behind the fallen-back ones: every image must still equal the oracle."""
    w, h = 150, 111
    p = oracle.full_params(2000, 0)
    rich = synth.stereo_batch(19, 2, w, h)
    simple = np.full((h, w), 60, np.uint8)
    simple[30:80, 40:110] = 200                              # one rectangle: four regions, a handful of chunks
    simple2 = np.full((h, w), 90, np.uint8)
    simple2[20:95, 20:35] = 10
    simple2[50:60, 60:140] = 250
    imgs = np.stack([simple, rich[0], rich[1], simple2, rich[2]])
    ex = ola.Lineextractor(0, 0.025, max_images=5)
    ex._params.orb.nlevels = 1                                # (the context also sizes an ORB pyramid: one level accepts a 150 x 111 image)
    p.orb.nlevels = 1
    want = [oracle.line_extract(im, p.line) for im in imgs]
    assert len(want[0]["kls"]) >= 4 and len(want[3]["kls"]) >= 4 and len(want[1]["kls"]) > 10
    for waves, rob, pool in [(8, 256, 300), (4, 128, 160), (16, 512, 560), (8, 256, 0)]:
        _set(ex, w, h, 5, waves, rob)
        _lib.check(_lib.lib().olf_debug_lsd_pool(ex._ctx.handle, pool), "olf_debug_lsd_pool")
        kls, desc, counts = ex.extract_batch(imgs)
        assert (_status(ex)[0] & (8 | 16)) == 0, (waves, rob, pool)
        for i in range(5):
            n = int(counts[i])
            assert n == len(want[i]["kls"]), (waves, pool, i, n, len(want[i]["kls"]))
            assert np.array_equal(kls[i, :n], want[i]["kls"]) and np.array_equal(desc[i, :n], want[i]["desc"]), (waves, pool, i)
    _lib.check(_lib.lib().olf_debug_lsd_pool(ex._ctx.handle, 0), "olf_debug_lsd_pool")
