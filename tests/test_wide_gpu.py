"""GPU parity of the capacity path (csrc/lsd_wide.hip): lsd_n_bins > 1024 and LSD working images of 2^22 pixels and more -- free YAML keys of the reference
(src/Config.cpp:268,274; Examples/PL/PL_KITTI00-02.yaml:110,116) whose sort key does not fit the fast path's 32-bit word.  The 64-bit seed-order kernel
against the real std::sort (oracle/line_oracle.cpp orc_std_sort_keys64) on arbitrary key arrays, and the whole line path against the oracle."""
import ctypes as C
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sorter():
    ex = ola.Lineextractor(0, 0.025, lsd_n_bins=4096, max_images=1)
    ctx = ex._context(1242, 375, 1)                 # a wide context: Ps = 1490 * 450 keys of capacity

    def run(keys, kthr=0xffffffff, depth=-1, full=False):
        keys = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros(max(len(keys), 1), np.uint32)
        n = C.c_int32()
        _lib.check(_lib.lib().olf_debug_seed_sort_wide(ctx.handle, _lib.ptr(keys), len(keys), int(kthr), int(depth), int(full), _lib.ptr(out), C.byref(n)),
                   "olf_debug_seed_sort_wide")
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
    k = np.clip(k, 0, nk - 1).astype(np.uint64)
    return (k << np.uint64(32)) | i.astype(np.uint64)


def test_wide_sort_equals_std_sort_on_random_arrays(oracle, sorter):
    """the order of the payloads the kernel lists = the order std::sort leaves (unstable: ties come out as libstdc++'s introsort leaves them)"""
    rng = np.random.default_rng(5)
    sizes = list(range(0, 36)) + [63, 64, 65, 127, 129, 255, 257, 511, 512, 513, 514, 600, 1023, 1025, 1537, 2049, 4097, 10000, 33333, 100001, 400000, 668561]
    for n in sizes:
        for mode in range(6):
            for nk in ((1, 3, 70000) if n > 5000 else (1, 2, 3, 17, 1024, 70000)):
                keys = _keys(rng, n, nk, mode)
                got = sorter(keys)
                want = (oracle.std_sort_keys64(keys) & np.uint64(0xffffffff)).astype(np.uint32)
                assert np.array_equal(got, want), (n, mode, nk, int(np.argmax(got != want)) if len(got) == len(want) else (len(got), len(want)))
    # the listed part is the prefix whose field is <= kthr; ranges that hold larger fields only may be left unsorted
    for n, kthr in [(5000, 10), (70000, 3), (70000, 5000), (2000, 0)]:
        keys = _keys(rng, n, 70000, 4)
        want = oracle.std_sort_keys64(keys)
        want = (want[(want >> np.uint64(32)) <= np.uint64(kthr)] & np.uint64(0xffffffff)).astype(np.uint32)
        assert np.array_equal(sorter(keys, kthr=kthr), want), (n, kthr)


def test_wide_sort_whole_word_order(oracle, sorter):
    """full = 1 (convention C.9 variant 0): distinct keys in ascending order of the whole word, from any input order"""
    rng = np.random.default_rng(6)
    for n in (0, 1, 2, 17, 100, 513, 5000, 70001, 300000):
        for nk in (1, 5, 3000):
            keys = _keys(rng, n, nk, 0)
            keys = keys[rng.permutation(n)]
            got = sorter(keys, full=True)
            assert np.array_equal(got, (np.sort(keys) & np.uint64(0xffffffff)).astype(np.uint32)), (n, nk)


def test_wide_sort_heap_sort_branch(oracle, sorter):
    """a forced depth limit sends every range that is still larger than 16 elements after `limit` partitions into libstdc++'s heap sort: ranges in LDS
    (<= 512 keys) and in memory"""
    rng = np.random.default_rng(7)
    for n in (17, 40, 64, 100, 511, 513, 700, 1537, 5000):
        for limit in (0, 1, 2, 5):
            for mode, nk in ((0, 70000), (0, 3), (4, 2048), (1, 50)):
                keys = _keys(rng, n, nk, mode)
                got = sorter(keys, depth=limit)
                want = (oracle.introsort_keys64(keys, limit) & np.uint64(0xffffffff)).astype(np.uint32)
                assert np.array_equal(got, want), (n, limit, mode, nk)


@pytest.mark.parametrize("seed_order", [1, 0])
@pytest.mark.parametrize("n_bins", [2048, 65536])
def test_line_extract_more_than_1024_bins(oracle, n_bins, seed_order):
    """KITTI size, lsd_n_bins beyond the 10-bit field of the fast path (Examples/PL/PL_KITTI00-02.yaml:116 is a free key): key lines and descriptors equal the
    oracle's under both seed-order conventions, a stereo pair in one call"""
    w, h = 1242, 375
    p = oracle.full_params(2000, 500)
    p.line.lsd_n_bins = n_bins
    p.line.conv_seed_order = seed_order
    ex = ola.Lineextractor(500, 0.025, lsd_n_bins=n_bins, conv_seed_order=seed_order, max_images=2)
    imgs = np.stack(synth.stereo_pair(21, w, h))
    kls, desc, counts = ex.extract_batch(imgs)
    for i in range(2):
        o = oracle.line_extract(imgs[i], p.line)
        c = int(counts[i])
        assert c > 100 and np.array_equal(kls[i, :c], o["kls"]) and np.array_equal(desc[i, :c], o["desc"]), (n_bins, seed_order, i)
    # ... and the bins matter: the default 1024 gives a different seed order, hence (on this scene) different lines
    p.line.lsd_n_bins = 1024
    assert not np.array_equal(oracle.line_extract(imgs[0], p.line)["kls"], kls[0, :int(counts[0])])


def test_line_extract_more_bins_with_refine(oracle):
    """lsd_refine = STD / ADV run inside the agent: its wide instantiations"""
    w, h = 640, 480
    for refine in (1, 2):
        p = oracle.full_params(1000, 200)
        p.line.lsd_n_bins = 3000
        p.line.lsd_refine = refine
        ex = ola.Lineextractor(200, 0.025, lsd_refine=refine, lsd_n_bins=3000)
        img = synth.stereo_pair(5, w, h)[0]
        gk, gd = ex(img)
        o = oracle.line_extract(img, p.line)
        assert len(gk) > 20 and np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), refine


@pytest.mark.parametrize("seed_order", [1, 0])
def test_line_extract_1080p_at_lsd_scale_2(oracle, seed_order):
    """1920 x 1080 at lsd_scale 2.0 (src/Config.cpp:268): a working image of 3840 x 2160 = 8.3 M pixels, beyond the 22 address bits of the fast path's key"""
    w, h = 1920, 1080
    p = oracle.full_params(4000, 1000)
    p.line.lsd_scale = 2.0
    p.line.conv_seed_order = seed_order
    ex = ola.Lineextractor(1000, 0.025, lsd_scale=2.0, conv_seed_order=seed_order, max_images=1)
    img = synth.stereo_pair(9, w, h)[0]
    gk, gd = ex(img)
    o = oracle.line_extract(img, p.line)
    assert len(gk) > 100 and np.array_equal(gk, o["kls"]) and np.array_equal(gd, o["desc"]), seed_order


def test_wide_keys_in_a_batch_context(oracle):
    """more than 2048 images per call (a batch context: the LSD work images live inside the key buffers, the pixel log is sized by a bound with a spill arena
    behind it) with 64-bit keys: 2080 small images at 4096 bins in one call, a sample of them against the oracle"""
    w, h, n = 320, 240, 2080
    p = oracle.full_params(300, 60)
    p.line.lsd_n_bins = 4096
    ex = ola.Lineextractor(60, 0.025, lsd_n_bins=4096, max_images=n)
    base = np.stack([synth.stereo_pair(40 + i, w, h)[i & 1] for i in range(16)])
    imgs = np.tile(base, (n // 16, 1, 1))[:n].copy()
    kls, desc, counts = ex.extract_batch(imgs)
    for i in (0, 1, 7, 15, 16, 1033, 2079):
        o = oracle.line_extract(imgs[i], p.line)
        c = int(counts[i])
        assert c > 5 and np.array_equal(kls[i, :c], o["kls"]) and np.array_equal(desc[i, :c], o["desc"]), i
