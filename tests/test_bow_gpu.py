"""GPU parity: DBoW2 vocabulary-tree transform (Frame::ComputeBoW) vs the CPU oracle."""
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import _lib

pytestmark = pytest.mark.gpu


def _features(oracle, parent, leaf, desc, n, seed):
    rng = np.random.default_rng(seed)
    f = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    words = np.flatnonzero(leaf)
    hit = rng.integers(0, len(words), n // 4)
    f[: n // 4] = desc[words[hit]]                                   # exact word hits (repeated words in one image)
    flip = rng.integers(0, 256, n // 4)
    f[np.arange(n // 4), flip // 8] ^= (1 << (flip % 8)).astype(np.uint8)   # ... one bit away
    return f


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 1), (10, 3, 0), (10, 3, 4), (3, 5, 2), (20, 2, 1), (1, 4, 2)])
def test_bow_words(oracle, k, L, levelsup):
    parent, leaf, desc, weight = oracle.random_vocabulary(k, L, 100 + k + L, tie_every=5, stop_every=11)
    V = oracle.OracleVoc.create(k, L, parent, leaf, desc, weight)
    G = ola.ORBVocabulary.from_arrays(k, L, parent, leaf, desc, weight)
    assert G.info()["n_words"] == int(leaf.sum()) == G.size()
    f = _features(oracle, parent, leaf, desc, 1500, 9)
    bow_o, fv_o = V.transform(f, levelsup)
    bow_g, fv_g = G.transform(f, levelsup)
    assert list(bow_g) == list(bow_o) and list(bow_g.values()) == list(bow_o.values())      # same words, same order, same doubles
    assert fv_g == fv_o and list(fv_g) == list(fv_o)


@pytest.mark.parametrize("scoring", range(6))
@pytest.mark.parametrize("weighting", range(4))
def test_bow_weighting_scoring(oracle, scoring, weighting):
    k, L = 6, 3
    parent, leaf, desc, weight = oracle.random_vocabulary(k, L, 77, stop_every=13)
    V = oracle.OracleVoc.create(k, L, parent, leaf, desc, weight, scoring, weighting)
    G = ola.ORBVocabulary.from_arrays(k, L, parent, leaf, desc, weight, scoring, weighting)
    f = _features(oracle, parent, leaf, desc, 800, scoring * 4 + weighting)
    bow_o, fv_o = V.transform(f, 2)
    bow_g, fv_g = G.transform(f, 2)
    assert list(bow_g.items()) == list(bow_o.items()) and fv_g == fv_o


def test_bow_text_file_and_empty(oracle, tmp_path):
    k, L = 5, 3
    parent, leaf, desc, weight = oracle.random_vocabulary(k, L, 3, tie_every=4)
    f = _features(oracle, parent, leaf, desc, 600, 1)
    f[-5:] = 0                                                      # all-zero features sit on the phantom node of a newline-terminated file
    for final_newline in (True, False):
        path = tmp_path / f"v{int(final_newline)}.txt"
        oracle.write_voc_text(path, k, L, parent, leaf, desc, weight, 0, 0, final_newline)
        G = ola.ORBVocabulary()
        assert G.loadFromTextFile(path)
        assert G.info()["n_nodes"] == len(parent) + int(final_newline)
        V = oracle.OracleVoc.load_text(path)
        bow_o, fv_o = V.transform(f, 1)
        bow_g, fv_g = G.transform(f, 1)
        assert list(bow_g.items()) == list(bow_o.items()) and fv_g == fv_o
        if final_newline:
            assert not any(i >= 595 for v in fv_g.values() for i in v)      # swallowed by the weight-0 phantom child of the root
    bad = tmp_path / "bad.txt"
    bad.write_text("hello world\n")
    assert not ola.ORBVocabulary().loadFromTextFile(bad)
    assert not ola.ORBVocabulary().loadFromTextFile(tmp_path / "missing.txt")
    E = ola.ORBVocabulary()
    assert E.empty() and E.transform(f) == ({}, {})
    G = ola.ORBVocabulary.from_arrays(k, L, parent, leaf, desc, weight)
    assert G.transform(np.zeros((0, 32), np.uint8)) == ({}, {})
    with pytest.raises(_lib.OlfError):
        ola.ORBVocabulary.from_arrays(k, L, parent[::-1].copy(), leaf, desc, weight)       # parent must precede child


def test_bow_full_size_orbvoc_shape(oracle):
    """k = 10, L = 6 (ORBvoc.txt's shape: 1,111,111 nodes, 10^6 words), 2000 descriptors, levelsup = 4 as in Frame::ComputeBoW"""
    k, L = 10, 6
    parent, leaf, desc, weight = oracle.random_vocabulary(k, L, 2024, tie_every=1001, stop_every=5003)
    assert len(parent) == 1111111
    V = oracle.OracleVoc.create(k, L, parent, leaf, desc, weight)
    G = ola.ORBVocabulary.from_arrays(k, L, parent, leaf, desc, weight)
    f = _features(oracle, parent, leaf, desc, 2000, 4)
    bow_o, fv_o = V.transform(f, 4)
    bow_g, fv_g = G.transform(f, 4)
    assert list(bow_g.items()) == list(bow_o.items()) and fv_g == fv_o
    assert len(fv_g) > 50 and max(fv_g) < 1 + 10 + 100 + 10 ** 6        # node ids at level L - 4 = 2 ... wherever creation order put them


def test_search_by_bow_with_transform(oracle):
    """SearchByBoW(KeyFrame, Frame) fed by real feature vectors from the GPU transform"""
    from test_search_gpu import _frames
    last, cur = _frames(oracle, seed=53)
    k, L = 10, 3
    parent, leaf, desc, weight = oracle.random_vocabulary(k, L, 8)
    # make the tree meaningful: centroids = actual descriptors of the key frame
    rng = np.random.default_rng(1)
    desc[1:] = last.mDescriptors[rng.integers(0, last.N, len(parent) - 1)]
    G = ola.ORBVocabulary.from_arrays(k, L, parent, leaf, desc, weight)
    V = oracle.OracleVoc.create(k, L, parent, leaf, desc, weight)
    _, last.mFeatVec = G.transform(last.mDescriptors, 2)
    _, cur.mFeatVec = G.transform(cur.mDescriptors, 2)
    assert last.mFeatVec == V.transform(last.mDescriptors, 2)[1]
    n_o, m_o = oracle.search_by_bow(last, cur, 0.8)
    n_g, m_g = ola.ORBmatcher(0.8, True).SearchByBoW(last, cur)
    assert n_g == n_o and np.array_equal(m_g, m_o) and n_g > 30


@pytest.mark.parametrize("k,L,levelsup,check", [(10, 3, 2, True), (10, 6, 4, True), (4, 4, 2, False)])
def test_search_by_bow_batch_on_device(oracle, k, L, levelsup, check):
    """olf_search_by_bow_batch_dev: Frame::ComputeBoW + ORBmatcher::SearchByBoW(prev frame as key frame, frame) for a batch of consecutive frames
    without leaving the device, against the oracle's SearchByBoW on the oracle's own feature vectors, pair by pair.  Some key-frame features hold
    no map point, some a bad one; a few words have weight 0 (their features are not in the FeatureVector)."""
    import torch
    import ctypes as C
    from orb_line_slam_amd import synth
    from orb_line_slam_amd._lib import FrameBuffers, check as chk, lib
    w, h, B = 1242, 375, 5
    p = oracle.full_params(2000, 100)
    ctx = _lib.Context(p, w, h, 2 * B)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity
    imgs = synth.stereo_batch(211, B, w, h)
    for i in range(1, B):                                            # consecutive frames: the same scene shifted by a few pixels
        imgs[2 * i] = np.roll(imgs[0], 2 * i, axis=1)
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(imgs).to(dev)
    kps = torch.zeros((2 * B, cap, 28), dtype=torch.uint8, device=dev); desc = torch.zeros((2 * B, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros(2 * B, dtype=torch.int32, device=dev)
    L_ = lib()
    s = torch.cuda.current_stream().cuda_stream
    chk(L_.olf_orb_extract_dev(ctx.handle, d_img.data_ptr(), 2 * B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), s), "olf_orb_extract_dev")
    torch.cuda.synchronize()
    kp_h = kps.cpu().numpy().view(ola.KEYPOINT_DTYPE).reshape(2 * B, cap); de_h = desc.cpu().numpy(); cn_h = counts.cpu().numpy()
    # vocabulary whose centroids are real descriptors; every 7th word gets weight 0
    parent, leaf, vdesc, weight = oracle.random_vocabulary(k, L, 31)
    rng = np.random.default_rng(5)
    src = np.concatenate([de_h[2 * i, :cn_h[2 * i]] for i in range(B)])
    vdesc[1:] = src[rng.integers(0, len(src), len(parent) - 1)]
    weight = weight.copy(); words = np.flatnonzero(leaf); weight[words[::7]] = 0.0
    G = ola.ORBVocabulary.from_arrays(k, L, parent, leaf, vdesc, weight, context=ctx)
    V = oracle.OracleVoc.create(k, L, parent, leaf, vdesc, weight)
    valid = rng.random((B, cap)) < 0.85; bad = rng.random((B, cap)) < 0.05
    d_valid = torch.from_numpy(valid.astype(np.uint8)).to(dev); d_bad = torch.from_numpy(bad.astype(np.uint8)).to(dev)
    m = torch.full((B - 1, cap), -7, dtype=torch.int32, device=dev); nm = torch.zeros(B - 1, dtype=torch.int32, device=dev)
    chk(L_.olf_search_by_bow_batch_dev(ctx.handle, G._h, B, 2, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), d_valid.data_ptr(), d_bad.data_ptr(),
                                       0.7, int(check), levelsup, m.data_ptr(), nm.data_ptr(), s), "olf_search_by_bow_batch_dev")
    torch.cuda.synchronize()
    m_h, nm_h = m.cpu().numpy(), nm.cpu().numpy()
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    views = []
    for i in range(B):
        n = int(cn_h[2 * i])
        v = ola.FrameView(kp_h[2 * i, :n], de_h[2 * i, :n], None, sf, bounds=(0.0, float(w), 0.0, float(h)))
        v.mp_valid, v.mp_bad = valid[i, :n].copy(), bad[i, :n].copy()
        _, v.mFeatVec = V.transform(v.mDescriptors, levelsup)
        views.append(v)
    total = 0
    for j in range(B - 1):
        n_o, m_o = oracle.search_by_bow(views[j], views[j + 1], 0.7, checkOri=check)
        nF = views[j + 1].N
        assert nm_h[j] == n_o, (j, nm_h[j], n_o)
        assert np.array_equal(m_h[j, :nF], m_o), (j, int(np.argmax(m_h[j, :nF] != m_o)))
        assert (m_h[j, nF:] == -1).all()
        total += n_o
    assert total > 100 * (B - 1)
    ctx.close()


@pytest.mark.parametrize("n", [0, 1, 3, 4, 1027, 2064 * 13 + 2])
def test_stereo_points_mask(n):
    """olf_stereo_points_mask_dev: mask[i] = mvDepth[i] > 0 (src/Tracking.cc:586-588), the d_mp_valid plane of the batched SearchByBoW; lengths that are
    not a multiple of the 4 floats one thread takes, -1 (no match), 0, NaN and denormal depths."""
    import torch
    from orb_line_slam_amd._lib import check as chk, lib
    ctx = _lib.Context(_lib.default_params(), 320, 240, 2)
    rng = np.random.default_rng(n)
    z = rng.choice(np.array([-1.0, 0.0, -0.0, np.nan, 1e-42, 0.5, 37.25, np.inf], np.float32), size=n + 8).astype(np.float32)
    dev = torch.device("cuda", 0)
    d_z = torch.from_numpy(z).to(dev); d_m = torch.full((n + 8,), 9, dtype=torch.uint8, device=dev)
    chk(lib().olf_stereo_points_mask_dev(ctx.handle, d_z.data_ptr(), n, d_m.data_ptr(), torch.cuda.current_stream().cuda_stream), "olf_stereo_points_mask_dev")
    torch.cuda.synchronize()
    m = d_m.cpu().numpy()
    assert np.array_equal(m[:n], (z[:n] > 0).astype(np.uint8)) and (m[n:] == 9).all()
    ctx.close()
