"""GPU parity: Hamming matchers and Frame::ComputeStereoMatches vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, matcher

pytestmark = pytest.mark.gpu


def _rand_desc(rng, n, dup=0):
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for k in range(dup):            # force exact ties / near duplicates
        i, j = rng.integers(0, n, 2)
        d[i] = d[j]
        d[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    return d


@pytest.mark.parametrize("n1,n2", [(500, 500), (1, 7), (200, 2), (37, 1), (5, 0), (2000, 1999), (300, 4096), (257, 4097), (700, 6001), (33, 31), (65, 33)])
def test_knn2_and_match(oracle, n1, n2):
    rng = np.random.default_rng(n1 * 1000 + n2)
    d1, d2 = _rand_desc(rng, n1, dup=n1 // 10), _rand_desc(rng, n2, dup=n2 // 10)
    if n2 >= 2:
        d2[1] = d2[0]               # equal rows: lower train index must win
        d1[0] = d2[0]
    i0, a, b = matcher.knn2(d1, d2)
    oi, oa, ob = oracle.knn2(d1, d2)
    assert np.array_equal(i0, oi) and np.array_equal(a, oa) and np.array_equal(b, ob)
    for nnr in (0.9, 0.75):
        for lr in (True, False):
            _, m = matcher.match(d1, d2, nnr, best_lr_matches=lr)
            assert np.array_equal(m, oracle.match_bf(d1, d2, nnr, lr))


def test_distance_matrix(oracle):
    rng = np.random.default_rng(5)
    d1, d2 = _rand_desc(rng, 33, 0), _rand_desc(rng, 65, 0)
    d1[0] = 0
    d2[0] = 255
    m = matcher.distance_matrix(d1, d2)
    assert m[0, 0] == 256
    for i in (0, 7, 32):
        for j in (0, 1, 64):
            assert m[i, j] == oracle.hamming256(d1[i], d2[j])
    assert ola.ORBmatcher.DescriptorDistance(d1[3], d2[4]) == oracle.hamming256(d1[3], d2[4])


@pytest.mark.parametrize("w,h,nf,fx,bf", [(640, 480, 1000, 435.2047, 47.9064), (1242, 375, 2000, 718.856, 386.1448)])
def test_stereo_points(oracle, w, h, nf, fx, bf):
    p = oracle.full_params(nf, 500, fx, bf)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=2)
    imgs = synth.stereo_batch(21, 2, w, h)
    f = fe.stereo_points(imgs)
    for i in range(2):
        o = oracle.stereo_points(imgs[2 * i], imgs[2 * i + 1], p)
        g = f.pair(i)
        assert np.array_equal(g["mvKeys"], o["kpsL"]) and np.array_equal(g["mvKeysRight"], o["kpsR"])
        assert np.array_equal(g["mDescriptors"], o["descL"]) and np.array_equal(g["mDescriptorsRight"], o["descR"])
        assert np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32))
        assert np.array_equal(g["mvDepth"].view(np.uint32), o["depth"].view(np.uint32))
        assert (o["uRight"] >= 0).sum() > 50   # the synthetic pair must actually produce stereo matches


def test_distinctive_descriptors(oracle):
    """MapPoint / MapLine::ComputeDistinctiveDescriptors over a batch of landmarks"""
    from orb_line_slam_amd import matcher
    rng = np.random.default_rng(21)
    obs = []
    for n in [0, 1, 2, 3, 4, 5, 7, 8, 16, 33, 64, 65, 130, 300, 1024] + rng.integers(1, 40, 400).tolist():
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        d = np.repeat(base[None], n, 0)
        noise = rng.random((n, 256)) < rng.uniform(0.0, 0.2)             # observations = one descriptor with a few bits flipped
        d ^= np.packbits(noise, axis=1)
        if n > 3 and rng.random() < 0.3:
            d[1] = d[0]                                                  # identical rows -> equal medians -> the first must win
        obs.append(d)
    got = matcher.ComputeDistinctiveDescriptors(obs)
    want = oracle.distinctive_descriptors(obs)
    assert np.array_equal(got, want) and got[0] == -1 and got[1] == 0
    with pytest.raises(Exception):
        matcher.ComputeDistinctiveDescriptors([np.zeros((1025, 32), np.uint8)])
    assert len(matcher.ComputeDistinctiveDescriptors([])) == 0
