"""GPU parity of the multi-wave LSD region growing (csrc/lsd_grow.hip): the detected key lines must not depend on how many waves grow an
image concurrently nor on the size of the reorder buffer, and must equal the oracle's sequential seed loop (oracle/line_oracle.cpp:121-148)."""
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib

pytestmark = pytest.mark.gpu


def _set(ex, w, h, n, waves, rob, groups=1, scatter=0):
    ctx = ex._context(w, h, n)
    _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, waves, rob), "olf_debug_lsd_waves")
    _lib.check(_lib.lib().olf_debug_lsd_groups(ctx.handle, groups), "olf_debug_lsd_groups")
    _lib.check(_lib.lib().olf_debug_lsd_scatter(ctx.handle, scatter), "olf_debug_lsd_scatter")


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
    # workgroups per image {1, 2, 4} (several CUs on one image: per-group reorder buffers over rank-interleaved 1024-seed windows) x the ten settings;
    # groups = 0 is the automatic choice (2 for a batch this small, lsd_grow_groups)
    for groups in (1, 2, 4, 0):
        for waves, rob in [(0, 0), (1, 128), (2, 128), (2, 256), (4, 256), (8, 512), (16, 512), (16, 128), (3, 256), (-1, 0)]:
            if groups != 1 and waves == 0:
                continue                                  # (the one-wave agent has no groups)
            _set(ex, w, h, 4, waves, rob, groups)
            kls, desc, counts = ex.extract_batch(imgs)
            assert (_status(ex)[0] & (8 | 16)) == 0, (groups, waves, rob, "capacity / watchdog flag")
            for i in range(4):
                n = int(counts[i])
                assert n == len(want[i]["kls"]), (groups, waves, rob, i, n, len(want[i]["kls"]))
                assert np.array_equal(kls[i, :n], want[i]["kls"]), (groups, waves, rob, i)
                assert np.array_equal(desc[i, :n], want[i]["desc"]), (groups, waves, rob, i)


def test_growth_noise_and_flat_images(oracle):
    """Pure noise (hundreds of thousands of tiny regions, the chunk pool's worst case that still fits) and a flat image (no seed at all)."""
    w, h = 320, 240
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    flat = np.full((h, w), 77, np.uint8)
    p = oracle.full_params(1000, 0)
    ex = ola.Lineextractor(0, 0.025, max_images=2)
    want = {id(img): oracle.line_extract(img, p.line) for img in (noise, flat)}
    for groups in (1, 2, 4):
        for waves in (16, 4, 1):
            _set(ex, w, h, 2, waves, 0, groups)
            for img in (noise, flat):
                gk, gd = ex(img)
                o = want[id(img)]
                assert np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), (groups, waves)


def test_growth_structured_images_across_workgroups(oracle):
    """Images whose regions are long and collide across the 1024-seed windows the groups deal out: a fan of lines through one point, a checkerboard
    (every edge has the same gradient magnitude: thousands of equal keys, regions of different windows meeting at every corner) and concentric rings."""
    w, h = 480, 360
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    fan = np.full((h, w), 40, np.uint8)
    ang = np.arctan2(yy - h / 2, xx - w / 2)
    fan[(np.floor(ang / (np.pi / 12)) % 2) == 0] = 210
    checker = (((xx // 24) + (yy // 24)) % 2 * 170 + 40).astype(np.uint8)
    rings = (np.sin(np.hypot(xx - w / 2, yy - h / 2) / 6.0) * 100 + 128).astype(np.uint8)
    p = oracle.full_params(1000, 0)
    ex = ola.Lineextractor(0, 0.025, max_images=3)
    imgs = np.stack([fan, checker, rings])
    want = [oracle.line_extract(im, p.line) for im in imgs]
    assert all(len(o["kls"]) > 8 for o in want)
    for groups in (1, 2, 4):
        for waves, rob in [(16, 512), (8, 256), (16, 128)]:
            _set(ex, w, h, 3, waves, rob, groups)
            kls, desc, counts = ex.extract_batch(imgs)
            assert (_status(ex)[0] & (8 | 16)) == 0, (groups, waves, rob)
            for i in range(3):
                n = int(counts[i])
                assert n == len(want[i]["kls"]), (groups, waves, rob, i, n, len(want[i]["kls"]))
                assert np.array_equal(kls[i, :n], want[i]["kls"]) and np.array_equal(desc[i, :n], want[i]["desc"]), (groups, waves, rob, i)


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


def test_large_noise_image_fits_the_sort_levels_and_falls_back_cleanly(oracle):
    """Every pixel defined (noise) on a 1280 x 720 image, one image per call: the grid-wide top levels of the seed sort leave the most ranges they ever can (their
    minimum range size adapts to the image so that the list holds them), the multi-wave growth's chunk pool overflows and the image is grown again by the one-wave
    agent -- no capacity / watchdog flag, result equal to the oracle."""
    w, h = 1280, 720
    rng = np.random.default_rng(17)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    p = oracle.full_params(1000, 0)
    ex = ola.Lineextractor(0, 0.025, max_images=1)
    o = oracle.line_extract(noise, p.line)
    for groups in (0, 1):
        _set(ex, w, h, 1, -1, 0, groups)
        gk, gd = ex(noise)
        assert (_status(ex)[0] & (8 | 16 | 64)) == 0, groups
        assert np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), groups


def test_cross_cu_hand_over_under_uneven_load():
    """tools/stress_mg.py: single-pair calls (two growth workgroups and eight sort workgroups per image -- owner words, watermarks and steal notices cross CU
    boundaries as agent-scope relaxed atomics) while another thread keeps every CU busy with 256-pair batches; every result must equal the unloaded first one.
    (An idle chip hides visibility bugs; this is the uneven-load form.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_mg.py"), "8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600).stdout.decode()
    assert "STRESS OK" in out, out[-1500:]


def _xcd_images(w=480, h=360):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    fan = np.full((h, w), 40, np.uint8)
    ang = np.arctan2(yy - h / 2, xx - w / 2)
    fan[(np.floor(ang / (np.pi / 12)) % 2) == 0] = 210
    checker = (((xx // 24) + (yy // 24)) % 2 * 170 + 40).astype(np.uint8)
    rings = (np.sin(np.hypot(xx - w / 2, yy - h / 2) / 6.0) * 100 + 128).astype(np.uint8)
    noise = np.random.default_rng(23).integers(0, 256, (h, w), dtype=np.uint8)
    return np.stack([fan, checker, rings, noise, synth.stereo_batch(31, 1, w, h)[0]])


def test_growth_groups_scattered_over_xcds(oracle):
    """"A speed matter only" held to a bit-exact result: with olf_debug_lsd_scatter the G groups of an image run on CONSECUTIVE workgroups --
    different XCDs under the hardware's round-robin placement, so owner words, watermarks and steal notices cross between L2s -- for G in {2, 4} on the structured
    images, noise and a synthetic scene, repeated (a visibility bug is a race: one pass proves little); every result equals the oracle."""
    w, h = 480, 360
    imgs = _xcd_images(w, h)
    p = oracle.full_params(1000, 0)
    want = [oracle.line_extract(im, p.line) for im in imgs]
    ex = ola.Lineextractor(0, 0.025, max_images=len(imgs))
    for scatter in (1, 0):
        for groups in (2, 4):
            for waves, rob in [(16, 512), (16, 128)]:
                _set(ex, w, h, len(imgs), waves, rob, groups, scatter)
                for rep in range(6):
                    kls, desc, counts = ex.extract_batch(imgs)
                    assert (_status(ex)[0] & (8 | 16)) == 0, (scatter, groups, waves, rob, rep)
                    for i in range(len(imgs)):
                        n = int(counts[i])
                        assert n == len(want[i]["kls"]), (scatter, groups, waves, rob, rep, i, n, len(want[i]["kls"]))
                        assert np.array_equal(kls[i, :n], want[i]["kls"]) and np.array_equal(desc[i, :n], want[i]["desc"]), (scatter, groups, waves, rob, rep, i)
    _set(ex, w, h, len(imgs), -1, 0, 0, 0)


def test_cross_xcd_hand_over_under_uneven_load():
    """tools/stress_mg.py with the groups of every image scattered over XCDs (OLF_LSD_SCATTER=1) while 256-pair batches keep every CU busy."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OLF_LSD_SCATTER="1")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_mg.py"), "8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600).stdout.decode()
    assert "STRESS OK" in out, out[-1500:]


def test_message_passing_litmus_between_cus_and_xcds():
    """tools/micro/mp_litmus.hip: the exact pattern the commit waves rely on -- relaxed agent-scope store (or atomicMin) ; s_waitcnt vmcnt(0) ; relaxed agent-scope
    store, read in the opposite order by a workgroup on another CU of the same XCD and on another XCD, under read-modify-write noise from 60 other workgroups -- must
    show no reader that saw the flag without the data it announces."""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "micro", "mp_litmus")
    if not os.path.exists(exe):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-o", exe, exe + ".hip"], check=True)
    out = subprocess.run([exe, "1000000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    d = json.loads(out.stdout.decode())
    assert out.returncode == 0 and d["ordering_holds"] is True, d
    real = [r for r in d["results"] if r and not r["variant"].startswith("control")]
    assert len(real) >= 5 and all(r["violations"] == 0 and r["distinct_flags_seen"] > 1000 for r in real), real
    assert any(r["placement"].startswith("different XCDs") for r in real)


def _stripes(w, h):
    yy, xx = np.mgrid[0:h, 0:w]
    a = (((xx // 5) % 2) * 150 + 50).astype(np.uint8)                 # vertical stripes: nearly every working pixel lies in a logged region
    b = (((xx + yy) // 7 % 2) * 160 + 40).astype(np.uint8)            # diagonal stripes
    return a, b


@pytest.mark.parametrize("refine", [0, 1, 2])
def test_pixel_log_spills_to_the_arena(oracle, refine):
    """The one-wave agent's pixel log is sized by a bound in batch contexts; an image that outgrows it is grown again by a second launch on a block of the spill
    arena (k_lsd_grow LOG_ROOM / `retry`, k_lsd_rect reads the block through LineGeom::spillOf).  olf_debug_lsd_log_cap forces the situation on a small
    context: caps from "every image overflows in its first large region" to "some overflow late", with lsd_refine NONE / STD / ADV (the refining agents keep ONE
    region in the log and read it back); every result equals the oracle, and no capacity flag is raised."""
    w, h = 480, 360
    sa, sb = _stripes(w, h)
    flat = np.full((h, w), 99, np.uint8)                              # (never spills: the arena holds three blocks for these four images)
    imgs = np.stack([synth.stereo_batch(41, 1, w, h)[0], sa, flat, sb])
    p = oracle.full_params(1000, 0)
    p.line.lsd_refine = refine
    want = [oracle.line_extract(im, p.line) for im in imgs]
    assert sum(len(o["kls"]) > 4 for o in want) == 3
    ex = ola.Lineextractor(0, 0.025, lsd_refine=refine, max_images=len(imgs))
    _set(ex, w, h, len(imgs), 0, 0)                                    # the one-wave agent
    L = _lib.lib()
    for cap in (40, 700, 20000, 0):
        _lib.check(L.olf_debug_lsd_log_cap(ex._ctx.handle, cap), "olf_debug_lsd_log_cap")
        kls, desc, counts = ex.extract_batch(imgs)
        assert (_status(ex)[0] & (8 | 16)) == 0, (refine, cap)
        for i in range(len(imgs)):
            n = int(counts[i])
            assert n == len(want[i]["kls"]), (refine, cap, i, n, len(want[i]["kls"]))
            assert np.array_equal(kls[i, :n], want[i]["kls"]) and np.array_equal(desc[i, :n], want[i]["desc"]), (refine, cap, i)


def test_exhausted_spill_arena_is_a_capacity_error():
    """More spilling images than the arena has blocks: the call must say OLF_ERR_CAPACITY (status flag 8), not write past a log."""
    w, h = 320, 240
    sa, _ = _stripes(w, h)
    imgs = np.stack([sa] * 8)
    ex = ola.Lineextractor(0, 0.025, max_images=8)
    _set(ex, w, h, 8, 0, 0)
    _lib.check(_lib.lib().olf_debug_lsd_log_cap(ex._ctx.handle, 64), "olf_debug_lsd_log_cap")      # arena of 6 blocks, 8 images spill
    with pytest.raises(_lib.OlfError):
        ex.extract_batch(imgs)
    _lib.check(_lib.lib().olf_debug_lsd_log_cap(ex._ctx.handle, 0), "olf_debug_lsd_log_cap")
    kls, desc, counts = ex.extract_batch(imgs)                        # the context works again with the full log
    assert counts.min() > 0 and np.array_equal(kls[0], kls[7])


def test_batch_context_small_log_and_aliased_work_images(oracle):
    """A context for more than 2048 images is a batch context: no owner words (one-wave agent only), pixel log for half of the pixels + spill arena, LSD blur
    and enlarged image inside the key buffers.  2056 small images -- scenes, stripes (most of their pixels lie in logged regions), noise -- in one call: spot-checked
    against the oracle, identical images must give identical results wherever they sit in the batch, and the stereo entry runs on the same context."""
    w, h = 150, 111
    p = oracle.full_params(300, 0)
    p.orb.nlevels = 1
    # (one image in 24 overflows the half-size log -- the arena holds blocks for one in 16)
    base = list(synth.stereo_batch(19, 11, w, h)) + [_stripes(w, h)[0]] + [np.random.default_rng(3).integers(0, 256, (h, w), dtype=np.uint8)]
    n = 2056
    imgs = np.stack([base[i % len(base)] for i in range(n)])
    want = [oracle.line_extract(im, p.line) for im in base]
    ex = ola.Lineextractor(0, 0.025, max_images=n)
    ex._params.orb.nlevels = 1
    ex._params.orb.nfeatures = 300
    kls, desc, counts = ex.extract_batch(imgs)
    assert (_status(ex)[0] & (8 | 16)) == 0
    assert int((counts[22::24] > 0).all())                           # the stripes, every one of them grown again on an arena block
    for i in list(range(2 * len(base))) + list(range(n - len(base) - 3, n)):
        o = want[i % len(base)]
        m = int(counts[i])
        assert m == len(o["kls"]), (i, m, len(o["kls"]))
        assert np.array_equal(kls[i, :m], o["kls"]) and np.array_equal(desc[i, :m], o["desc"]), i
    for j in range(len(base)):
        same = counts[j::len(base)]
        assert (same == same[0]).all(), j
    with pytest.raises(_lib.OlfError):
        ex.debug_scaled(0)                                             # a batch context does not keep the enlarged image
