"""GPU parity of the std::sort seed order (convention C.9 variant 1, csrc/lsd_seedsort.hip): the kernel that replays libstdc++'s introsort
against the real std::sort (oracle/line_oracle.cpp orc_std_sort_keys), on arbitrary key arrays and through the whole line path."""
import ctypes as C
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sorter():
    ex = ola.Lineextractor(0, 0.025, max_images=1)
    ctx = ex._context(1242, 375, 1)                 # Ps = 1490 * 450 keys of capacity

    def run(keys, kthr=1023, depth=-1):
        keys = np.ascontiguousarray(keys, np.uint32)
        out = np.zeros(max(len(keys), 1), np.uint32)
        n = C.c_int32()
        _lib.check(_lib.lib().olf_debug_seed_sort(ctx.handle, _lib.ptr(keys), len(keys), int(kthr), int(depth), _lib.ptr(out), C.byref(n)), "olf_debug_seed_sort")
        return out[:n.value].copy()
    yield run
    ex._ctx.close()


def _keys(rng, n, nk, mode):
    i = np.arange(n)
    if mode == 0: k = rng.integers(0, nk, n)
    elif mode == 1: k = i * nk // max(n, 1)                                 # ascending
    elif mode == 2: k = nk - 1 - i * nk // max(n, 1)                        # descending
    elif mode == 3: k = np.where(i % 2 == 1, rng.integers(0, nk, n), 0)     # every other key minimal
    elif mode == 4: k = np.minimum(rng.geometric(0.02, n), nk) - 1          # skewed like gradient bins: most keys in a few values
    else: k = (rng.integers(0, nk, n) * (rng.random(n) < 0.2)).astype(np.int64) + (nk - 1) * (rng.random(n) < 0.6)
    k = np.clip(k, 0, 1023).astype(np.uint32)
    return (k << 22) | i.astype(np.uint32)


def test_sort_kernel_equals_std_sort_on_random_arrays(oracle, sorter):
    rng = np.random.default_rng(2)
    sizes = list(range(0, 40)) + [63, 64, 65, 66, 127, 128, 129, 130, 191, 192, 193, 255, 256, 257, 300, 511, 513, 1000, 1535, 1536, 1537, 1538,
                                  1600, 2047, 2049, 3000, 4097, 10000, 33333, 100001]
    for n in sizes:
        for mode in range(6):
            for nk in (1, 2, 3, 17, 1024):
                keys = _keys(rng, n, nk, mode)
                got = sorter(keys)
                want = oracle.std_sort_keys(keys)
                assert np.array_equal(got, want), (n, mode, nk, int(np.argmax(got != want)) if len(got) == len(want) else (len(got), len(want)))
    # the listed part is the prefix of the sorted array whose field is <= kthr, and ranges of larger fields may be left unsorted
    for n, kthr in [(5000, 10), (70000, 3), (70000, 500), (2000, 0)]:
        keys = _keys(rng, n, 1024, 4)
        want = oracle.std_sort_keys(keys)
        assert np.array_equal(sorter(keys, kthr=kthr), want[(want >> 22) <= kthr]), (n, kthr)


def test_sort_kernel_variants_agree(oracle):
    """one wave per image (the batch kernel), 4 / 8 cooperating waves per image (few images; ranges handed from wave to wave through a stack
    in LDS, larger blocks) and groups of 4 / 8 images per workgroup whose waves take over each other's ranges give the same list: arrays that
    keep every wave busy, and the line path on seven KITTI-sized images (a group that is not full, an image without any seed) under each variant"""
    ex = ola.Lineextractor(500, 0.025, max_images=7)
    ctx = ex._context(1242, 375, 7)
    rng = np.random.default_rng(8)
    cases = [_keys(rng, n, nk, mode) for n, nk, mode in [(400000, 1024, 4), (668561, 1024, 4), (100000, 1024, 0), (250000, 7, 0), (70000, 1, 0), (5000, 1024, 5)]]
    p = oracle.full_params(2000, 500)
    imgs = np.concatenate([synth.stereo_batch(17, 3, 1242, 375), np.full((1, 375, 1242), 90, np.uint8)])
    imgs = imgs[[0, 1, 6, 2, 3, 4, 5]]
    want_lines = [oracle.line_extract(im, p.line) for im in imgs]
    assert len(want_lines[2]["kls"]) == 0 and len(want_lines[0]["kls"]) > 100
    for mode in (0, 1, 2, 3, 4, 5):
        _lib.check(_lib.lib().olf_debug_seed_sort_mode(ctx.handle, mode), "olf_debug_seed_sort_mode")
        for keys in cases:
            out = np.zeros(len(keys), np.uint32); n = C.c_int32()
            _lib.check(_lib.lib().olf_debug_seed_sort(ctx.handle, _lib.ptr(keys), len(keys), 1023, -1, _lib.ptr(out), C.byref(n)), "olf_debug_seed_sort")
            assert n.value == len(keys) and np.array_equal(out, oracle.std_sort_keys(keys)), (mode, len(keys))
        kls, desc, counts = ex.extract_batch(imgs)
        for i in range(len(imgs)):
            c = int(counts[i])
            assert np.array_equal(kls[i, :c], want_lines[i]["kls"]) and np.array_equal(desc[i, :c], want_lines[i]["desc"]), (mode, i)
    _lib.check(_lib.lib().olf_debug_seed_sort_mode(ctx.handle, -1), "olf_debug_seed_sort_mode")


def test_sort_kernel_heap_sort_branch(oracle, sorter):
    """a forced depth limit sends every range that is still larger than 16 elements after `limit` partitions into libstdc++'s heap sort"""
    rng = np.random.default_rng(3)
    for n in (17, 40, 64, 100, 700, 1536, 1537, 5000, 40000):
        for limit in (0, 1, 2, 5):
            for mode, nk in ((0, 1024), (0, 3), (4, 1024), (1, 50)):
                keys = _keys(rng, n, nk, mode)
                got = sorter(keys, depth=limit)
                want = oracle.introsort_keys(keys, limit)
                assert np.array_equal(got, want), (n, limit, mode, nk)


@pytest.mark.parametrize("w,h,nl", [(640, 480, 200), (1242, 375, 500), (752, 480, 0)])
def test_line_extract_std_sort_seed_order(oracle, w, h, nl):
    """C2 / C3 / C4 sizes through the line path with conv_seed_order = 1: key lines and descriptors equal the oracle's, which calls the real
    std::sort on the {pixel, bin} vector"""
    p = oracle.full_params(2000, nl)
    p.line.conv_seed_order = 1
    ex = ola.Lineextractor(nl, 0.025, conv_seed_order=1)
    for seed in (3, 4):
        left, right = synth.stereo_pair(seed, w, h)
        for img in (left, right):
            gk, gd = ex(img)
            o = oracle.line_extract(img, p.line)
            assert np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), (seed,)
            assert len(gk) > 20


def test_std_sort_seed_order_structured_and_noise(oracle):
    p = oracle.full_params(1000, 0)
    p.line.conv_seed_order = 1
    ex = ola.Lineextractor(0, 0.025, conv_seed_order=1)
    h, w = 240, 320
    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(0)
    images = {
        "noise": rng.integers(0, 256, (h, w), dtype=np.uint8),
        "checker": ((((x // 16) + (y // 16)) % 2) * 255).astype(np.uint8),
        "flat": np.full((h, w), 128, np.uint8),
        "edge": np.where(x >= 160, 255, 0).astype(np.uint8),
        "rings": np.clip(np.rint(np.abs((np.hypot(x - w / 2, y - h / 2) % 64) - 32) * 7.5), 0, 255).astype(np.uint8),
    }
    for name, img in images.items():
        gk, gd = ex(img)
        o = oracle.line_extract(img, p.line)
        assert np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), name
    # multi-wave growth kernels read the same seed list
    ctx = ex._context(w, h, 1)
    for waves in (16, 4, 0):
        _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, waves, 0), "olf_debug_lsd_waves")
        gk, gd = ex(images["rings"])
        o = oracle.line_extract(images["rings"], p.line)
        assert np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), waves


def test_sort_paths_by_batch_size(oracle):
    """the first partitions run as grid-wide kernels for up to 64 images per call and inside the per-image workgroup beyond: 6, 64, 65 and 130
    images of one size through the line path, every image against the oracle"""
    w, h, distinct = 480, 320, 13
    base = synth.stereo_batch(900, 7, w, h)[:distinct]
    p = oracle.full_params(1000, 300)
    want = [oracle.line_extract(im, p.line) for im in base]
    for n in (6, 64, 65, 130):
        imgs = np.tile(base, (n // distinct + 1, 1, 1))[:n]
        ex = ola.Lineextractor(300, 0.025, max_images=n)
        kls, desc, counts = ex.extract_batch(imgs)
        for i in range(n):
            o = want[i % distinct]
            c = int(counts[i])
            assert c == len(o["kls"]) and np.array_equal(kls[i, :c], o["kls"]) and np.array_equal(desc[i, :c], o["desc"]), (n, i)
