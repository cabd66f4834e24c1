"""GPU parity: ORBmatcher::SearchByProjection(Frame, Frame) and SearchByBoW(KeyFrame, Frame) -- host candidate lists and greedy
resolution as in the reference, every descriptor distance from the GPU -- against the CPU oracle."""
import copy
import numpy as np
import pytest
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth

pytestmark = pytest.mark.gpu


def _frames(oracle, w=1242, h=375, seed=41):
    """two consecutive 'frames': the left images of two seeds' stereo pairs; map points from the stereo depth of frame 0"""
    p = oracle.full_params(2000, 500)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=2)
    imgs = synth.stereo_batch(seed, 2, w, h)
    imgs[2:] = np.roll(imgs[:2], 3, axis=2)            # frame 1 = frame 0 shifted by 3 px: a small known motion
    f = fe.stereo_points(imgs)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    views = []
    for i in range(2):
        g = f.pair(i)
        views.append(ola.FrameView(g["mvKeys"], g["mDescriptors"], g["mvuRight"], sf, bounds=(0.0, float(w), 0.0, float(h))))
    last, cur = views
    # back-project the stereo points of the last frame (identity pose) as its map points
    depth = f.pair(0)["mvDepth"]
    ok = depth > 0
    z = np.where(ok, depth, 1).astype(np.float32)
    last.mp_valid = ok.copy()
    last.mp_world = np.stack([(last.mvKeysUn["x"] - last.cx) * z / last.fx, (last.mvKeysUn["y"] - last.cy) * z / last.fy, z], 1).astype(np.float32)
    last.mp_desc = last.mDescriptors.copy()
    last.mp_obs = ok.copy()
    last.mvbOutlier[::17] = True
    return last, cur


@pytest.mark.parametrize("th,bMono", [(7, False), (14, False), (15, True)])
def test_search_by_projection(oracle, th, bMono):
    last, cur = _frames(oracle)
    cur.mTcw = np.eye(4, dtype=np.float32)
    cur.mTcw[0, 3] = 0.02                              # small predicted translation
    cur_o = copy.deepcopy(cur)
    n_o, m_o = oracle.search_by_projection(cur_o, last, th, bMono)
    n_g, m_g = ola.ORBmatcher(0.9, True).SearchByProjection(cur, last, th, bMono)
    assert n_g == n_o and np.array_equal(m_g, m_o)
    assert n_g > 100
    n2, m2 = ola.ORBmatcher(0.9, False).SearchByProjection(copy.deepcopy(cur_o), last, th, bMono)
    n2o, m2o = oracle.search_by_projection(cur_o, last, th, bMono, checkOri=False)
    assert n2 == n2o and np.array_equal(m2, m2o)


@pytest.mark.parametrize("th,bMono,check", [(7, False, True), (14, False, False), (15, True, True)])
def test_search_by_projection_match12(oracle, th, bMono, check):
    """The overload Tracking::TrackWithMotionModelWithLine calls (src/ORBmatcher.cc:1474-1618, src/Tracking.cc:1296,1302): same search, plus the
    map<int,int> match12.  Temporal points (map points without observations, Tracking::UpdateLastFrame) let a current feature be matched twice:
    mvpMapPoints[i2] then keeps the LAST last-frame point while match12.insert keeps the FIRST -- the last frame's points are listed twice
    here (second copy with perturbed descriptors and Observations() == 0 everywhere) so that this happens many times."""
    last, cur = _frames(oracle)
    n0 = last.N
    two = copy.deepcopy(last)
    for name in ("mvKeysUn", "mDescriptors", "mvuRight", "mp_valid", "mp_world", "mp_desc", "mp_obs", "mp_bad", "mvbOutlier"):
        a = getattr(last, name)
        setattr(two, name, np.concatenate([a, a]))
    two.N, two.mvKeys = 2 * n0, two.mvKeysUn
    two.mp_obs[:] = False                                    # every map point of the last frame is a temporal one
    rng = np.random.default_rng(9)
    flip = rng.integers(0, 256, (n0, 32)).astype(np.uint8) & rng.integers(0, 256, (n0, 32)).astype(np.uint8) & rng.integers(0, 256, (n0, 32)).astype(np.uint8) & 0x11
    two.mp_desc[n0:] ^= flip                                 # a few bits: the second copy still matches the same current feature
    two.mvKeysUn["angle"][n0:] = (two.mvKeysUn["angle"][n0:] + rng.choice([0.0, 0.0, 120.0], n0)).astype(np.float32) % 360   # some land in another rotation bin
    cur.mTcw = np.eye(4, dtype=np.float32)
    cur.mTcw[0, 3] = 0.02
    cur_o = copy.deepcopy(cur)
    n_o, m_o, pairs_o, valid_o = oracle.search_by_projection_match12(cur_o, two, th, bMono, checkOri=check)
    m12 = {7: 7}
    n_g, m_g = ola.ORBmatcher(0.9, check).SearchByProjection(cur, two, th, bMono, m12)
    assert n_g == n_o and np.array_equal(m_g, m_o)
    assert list(m12.items()) == pairs_o                      # same keys, same values, same (ascending) order
    assert np.array_equal(cur.mp_valid.astype(np.uint8), valid_o)
    differ = sum(1 for k, v in pairs_o if m_o[k] != v)
    assert differ > 20, differ                               # insert-keeps-first really differs from the last assignment
    assert n_g != len(pairs_o) or not check                  # nmatches counts both assignments of a twice-matched feature


@pytest.mark.parametrize("window,ratio", [(100, 0.9), (30, 0.7), (10, 0.9)])
def test_search_for_initialization(oracle, window, ratio):
    """SearchForInitialization (monocular initialisation, src/ORBmatcher.cc:407-522): level-0 key points only, greedy with re-assignment
    (vMatchedDistance / vnMatches21), vbPrevMatched updated in place"""
    f1, f2 = _frames(oracle, seed=47)
    prev = np.ascontiguousarray(np.stack([f1.mvKeysUn["x"], f1.mvKeysUn["y"]], axis=1).astype(np.float32))
    for check in (True, False):
        n_o, m_o, pm_o = oracle.search_for_initialization(f1, f2, prev, window, ratio, checkOri=check)
        pm_g = prev.copy()
        n_g, m_g = ola.ORBmatcher(ratio, check).SearchForInitialization(f1, f2, pm_g, window)
        assert n_g == n_o and np.array_equal(m_g, m_o) and np.array_equal(pm_g, pm_o)
        assert n_g == int((m_g >= 0).sum())
    assert n_g > (20 if window >= 30 else 0)
    # second call seeded with the updated positions, as Tracking::MonocularInitialization does frame after frame
    n_o2, m_o2, pm_o2 = oracle.search_for_initialization(f1, f2, pm_o, window, ratio, checkOri=False)
    pm_g2 = pm_g.copy()
    n_g2, m_g2 = ola.ORBmatcher(ratio, False).SearchForInitialization(f1, f2, pm_g2, window)
    assert n_g2 == n_o2 and np.array_equal(m_g2, m_o2) and np.array_equal(pm_g2, pm_o2)


def test_search_by_bow(oracle):
    last, cur = _frames(oracle, seed=43)
    # synthetic feature vectors (the vocabulary blobs are missing from the reference checkout): node = coarse image cell
    def fv(v):
        d = {}
        for i in range(v.N):
            d.setdefault(int(v.mvKeysUn["x"][i] // 80) * 10 + int(v.mvKeysUn["y"][i] // 80), []).append(i)
        return d
    last.mFeatVec, cur.mFeatVec = fv(last), fv(cur)
    last.mp_bad[::29] = True
    for ratio in (0.7, 0.9):
        n_o, m_o = oracle.search_by_bow(last, cur, ratio)
        n_g, m_g = ola.ORBmatcher(ratio, True).SearchByBoW(last, cur)
        assert n_g == n_o and np.array_equal(m_g, m_o)
    assert n_g > 50


def test_candidate_distances_edge_cases(oracle):
    from orb_line_slam_amd import matcher
    rng = np.random.default_rng(3)
    q, t = rng.integers(0, 256, (5, 32), dtype=np.uint8), rng.integers(0, 256, (9, 32), dtype=np.uint8)
    lists = [[0, 8, 3], [], [2], [9, -1, 4], [1] * 6]                     # empty list, out-of-range indices, repeats
    d = matcher._candidate_distances(q, lists, t)
    assert [len(x) for x in d] == [3, 0, 1, 3, 6]
    assert d[0][1] == oracle.hamming256(q[0], t[8]) and d[3][0] == 0xffff and d[3][1] == 0xffff and d[3][2] == oracle.hamming256(q[3], t[4])
    assert matcher._candidate_distances(np.zeros((0, 32), np.uint8), [], t) == []


@pytest.mark.parametrize("th,ratio", [(1.0, 0.8), (3.0, 0.8), (5.0, 0.6)])
def test_search_local_map(oracle, th, ratio):
    """SearchByProjection(Frame, vector<MapPoint*>, th): the local map = the last frame's stereo points, projected with a small offset"""
    last, cur = _frames(oracle, seed=47)
    rng = np.random.default_rng(5)
    sel = np.flatnonzero(last.mp_valid)
    n = len(sel)
    px = (last.mvKeysUn["x"][sel] + 3 + rng.normal(0, 1.0, n)).astype(np.float32)      # frame 1 is frame 0 shifted by 3 px
    py = (last.mvKeysUn["y"][sel] + rng.normal(0, 1.0, n)).astype(np.float32)
    pxr = (px - (last.mvKeysUn["x"][sel] - np.where(last.mvuRight[sel] > 0, last.mvuRight[sel], 0))).astype(np.float32)
    lvl = np.clip(last.mvKeysUn["octave"][sel] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    cos = rng.choice(np.array([0.9, 0.9979, 0.998, 0.9981, 1.0], np.float32), n)
    mp = ola.MapPointView(last.mDescriptors[sel], px, py, pxr, lvl, cos, mbTrackInView=rng.random(n) > 0.1, isBad=rng.random(n) < 0.05,
                          obs=rng.random(n) > 0.3)
    cur.mp_valid[::7] = True; cur.mp_obs[::14] = True       # some features already carry a map point (with / without observations)
    cur_o = copy.deepcopy(cur)
    n_o, m_o = oracle.search_local_map(cur_o, mp, th, ratio)
    n_g, m_g = ola.ORBmatcher(ratio, True).SearchByProjection(cur, mp, th)
    assert n_g == n_o and np.array_equal(m_g, m_o)
    assert n_g > 200


def _as_kf(view):
    kf = ola.KeyFrameView(view.mvKeysUn, view.mDescriptors, view.mvuRight, view.mvScaleFactors,
                          bounds=(float(view.mnMinX), float(view.mnMaxX), float(view.mnMinY), float(view.mnMaxY)))
    for a in ("mp_valid", "mp_world", "mp_desc", "mp_obs", "mp_bad", "mvbOutlier", "mFeatVec"):
        setattr(kf, a, copy.deepcopy(getattr(view, a)))
    return kf


def test_search_by_bow_kf_kf(oracle):
    """SearchByBoW(KeyFrame, KeyFrame) (loop closing): both sides carry map points; strict TH_LOW, vbMatched2"""
    last, cur = _frames(oracle, seed=59)
    def fv(v):
        d = {}
        for i in range(v.N):
            d.setdefault(int(v.mvKeysUn["x"][i] // 90) * 7 + int(v.mvKeysUn["y"][i] // 90), []).append(i)
        return d
    kf1, kf2 = _as_kf(last), _as_kf(cur)
    kf1.mFeatVec, kf2.mFeatVec = fv(kf1), fv(kf2)
    rng = np.random.default_rng(2)
    kf2.mp_valid = rng.random(kf2.N) > 0.2; kf2.mp_bad = rng.random(kf2.N) < 0.05; kf1.mp_bad[::31] = True
    for ratio in (0.75, 0.9):
        n_o, m_o = oracle.search_by_bow_kf(kf1, kf2, ratio)
        n_g, m_g = ola.ORBmatcher(ratio, True).SearchByBoW(kf1, kf2)
        assert n_g == n_o and np.array_equal(m_g, m_o)
    assert n_g > 40
    n2o, m2o = oracle.search_by_bow_kf(kf1, kf2, 0.9, checkOri=False)
    n2g, m2g = ola.ORBmatcher(0.9, False).SearchByBoW(kf1, kf2)
    assert n2g == n2o and np.array_equal(m2g, m2o)


@pytest.mark.parametrize("th,ORBdist", [(10, 100), (3, 64)])
def test_search_by_projection_keyframe(oracle, th, ORBdist):
    """SearchByProjection(Frame, KeyFrame, sAlreadyFound, th, ORBdist) (relocalisation): predicted scale level from the viewing distance"""
    last, cur = _frames(oracle, seed=61)
    kf = _as_kf(last)
    rng = np.random.default_rng(7)
    d = np.linalg.norm(kf.mp_world, axis=1).astype(np.float32)
    lvl = kf.mvKeysUn["octave"].astype(np.float32)
    kf.mp_maxd = (d * np.float32(1.2) ** lvl * rng.uniform(0.9, 1.1, kf.N)).astype(np.float32)        # ~ dist * levelScaleFactor
    kf.mp_mind = (kf.mp_maxd / np.float32(1.2) ** 7).astype(np.float32)
    found = rng.random(kf.N) < 0.1
    cur.mTcw = np.eye(4, dtype=np.float32); cur.mTcw[0, 3] = 0.015; cur.mTcw[2, 3] = -0.05
    cur.mp_valid[::9] = True
    cur_o = copy.deepcopy(cur)
    n_o, m_o = oracle.search_by_projection_kf(cur_o, kf, found, th, ORBdist)
    n_g, m_g = ola.ORBmatcher(0.9, True).SearchByProjection(cur, kf, found, th, ORBdist)
    assert n_g == n_o and np.array_equal(m_g, m_o)
    assert n_g > 100


def _fundamental_for_shift(kf, t):
    """F12 for two identical-orientation cameras, camera 2 displaced by t: nearly horizontal epipolar lines for t ~ (tx, 0, small)"""
    K = np.array([[kf.fx, 0, kf.cx], [0, kf.fy, kf.cy], [0, 0, 1]], np.float64)
    tcross = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]], np.float64)
    Kinv = np.linalg.inv(K)
    return (Kinv.T @ tcross @ Kinv).astype(np.float32)      # used as x1^T F12 x2 by CheckDistEpipolarLine


@pytest.mark.parametrize("bOnlyStereo", [False, True])
def test_search_for_triangulation(oracle, bOnlyStereo):
    last, cur = _frames(oracle, seed=67)
    kf1, kf2 = _as_kf(last), _as_kf(cur)
    def fv(v):
        d = {}
        for i in range(v.N):
            d.setdefault(int(v.mvKeysUn["y"][i] // 40), []).append(i)
        return d
    kf1.mFeatVec, kf2.mFeatVec = fv(kf1), fv(kf2)
    rng = np.random.default_rng(11)
    kf1.mp_valid = rng.random(kf1.N) < 0.4; kf2.mp_valid = rng.random(kf2.N) < 0.4      # only features WITHOUT a map point are searched
    kf2.mTcw = np.eye(4, dtype=np.float32); kf2.mTcw[0, 3] = -0.3; kf2.mTcw[2, 3] = 0.004      # finite epipole, far to the side
    F12 = _fundamental_for_shift(kf1, (0.3, 0.0, -0.004))
    Cw = np.zeros(3, np.float32)
    n_o, p_o = oracle.search_for_triangulation(kf1, kf2, F12, bOnlyStereo, Cw)
    n_g, p_g = ola.ORBmatcher(0.6, True).SearchForTriangulation(kf1, kf2, F12, bOnlyStereo, Cw)
    assert n_g == n_o and p_g == p_o
    assert n_g > 30
    n2o, p2o = oracle.search_for_triangulation(kf1, kf2, F12, bOnlyStereo, Cw, checkOri=False)
    n2g, p2g = ola.ORBmatcher(0.6, False).SearchForTriangulation(kf1, kf2, F12, bOnlyStereo, Cw)
    assert n2g == n2o and p2g == p2o


@pytest.mark.parametrize("th", [3.0, 6.0])
def test_fuse_search(oracle, th):
    last, cur = _frames(oracle, seed=71)
    kf = _as_kf(cur)
    sel = np.flatnonzero(last.mp_valid)
    rng = np.random.default_rng(13)
    world = last.mp_world[sel].copy(); world[:, 0] += np.float32(3 * 0.54 / 718.856) * world[:, 2]      # frame 1 = frame 0 shifted by 3 px
    d = np.linalg.norm(world, axis=1).astype(np.float32)
    lvl = last.mvKeysUn["octave"][sel].astype(np.float32)
    maxd = (d * np.float32(1.2) ** lvl * rng.uniform(0.95, 1.05, len(sel))).astype(np.float32)
    normal = (-world / d[:, None] + rng.normal(0, 0.3, world.shape)).astype(np.float32)                # some beyond the 60 degree gate
    mp = ola.MapPointGeom(world, -normal, maxd, maxd / np.float32(1.2) ** 7, last.mDescriptors[sel], skip=rng.random(len(sel)) < 0.1)
    Ow = np.zeros(3, np.float32)
    bi_o, bd_o = oracle.fuse_search(kf, mp, th, Ow)
    bi_g, bd_g = ola.ORBmatcher(0.6, True).FuseSearch(kf, mp, th, Ow)
    assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o)
    assert (bi_g >= 0).sum() > 200 and ((bd_g <= 50) & (bi_g >= 0)).sum() > 100


def _rot(ax, ay, az):
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


@pytest.mark.parametrize("th,scale", [(4.0, 1.0), (10.0, 1.03)])
def test_fuse_search_sim3(oracle, th, scale):
    """Fuse(pKF, Scw, vpPoints, th, vpReplacePoint), the loop-correction variant: pose given as a similarity, no stereo / chi-square gate"""
    last, cur = _frames(oracle, seed=83)
    kf = _as_kf(cur)
    sel = np.flatnonzero(last.mp_valid)
    rng = np.random.default_rng(29)
    world = last.mp_world[sel].copy(); world[:, 0] += np.float32(3 * 0.54 / 718.856) * world[:, 2]
    d = np.linalg.norm(world, axis=1).astype(np.float32)
    lvl = last.mvKeysUn["octave"][sel].astype(np.float32)
    maxd = (d * np.float32(1.2) ** lvl * rng.uniform(0.95, 1.05, len(sel))).astype(np.float32)
    normal = (world / d[:, None] + rng.normal(0, 0.3, world.shape)).astype(np.float32)
    mp = ola.MapPointGeom(world, normal, maxd, maxd / np.float32(1.2) ** 7, last.mDescriptors[sel], skip=rng.random(len(sel)) < 0.1)
    # world frame = camera frame moved by a small rigid motion and scaled: Scw = s [R | t]
    R, t = _rot(0.002, -0.003, 0.001), np.array([0.01, -0.004, 0.02], np.float32)
    Scw = np.eye(4, dtype=np.float32); Scw[:3, :3] = np.float32(scale) * R; Scw[:3, 3] = np.float32(scale) * t
    mp.world = ((mp.world - t) @ R).astype(np.float32)                       # so that Rcw * p + tcw lands on the original points
    Ro, to, Oo = oracle.sim3_decompose(Scw)
    Rg, tg, Og = ola.matcher.Sim3Decompose(Scw)
    assert np.array_equal(Ro.view(np.uint32), Rg.view(np.uint32)) and np.array_equal(to.view(np.uint32), tg.view(np.uint32))
    assert np.array_equal(Oo.view(np.uint32), Og.view(np.uint32))
    bi_o, bd_o = oracle.fuse_search_sim3(kf, Scw, mp, th)
    bi_g, bd_g = ola.ORBmatcher(0.6, True).FuseSearchSim3(kf, Scw, mp, th)
    assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o)
    assert (bi_g >= 0).sum() > 200 and ((bd_g <= 50) & (bi_g >= 0)).sum() > 100


@pytest.mark.parametrize("th,scale", [(10, 1.0), (4, 1.03)])
def test_search_by_projection_sim3(oracle, th, scale):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:292-405, LoopClosing::ComputeSim3): greedy -- a key point taken by
    an earlier point (or matched before the call) is passed over by the later ones.  Every map point is listed twice (the second copy with a
    few descriptor bits flipped), so the second copy has to settle for another key point or for none."""
    last, cur = _frames(oracle, seed=83)
    kf = _as_kf(cur)
    sel = np.flatnonzero(last.mp_valid)
    rng = np.random.default_rng(37)
    world = last.mp_world[sel].copy(); world[:, 0] += np.float32(3 * 0.54 / 718.856) * world[:, 2]
    d = np.linalg.norm(world, axis=1).astype(np.float32)
    lvl = last.mvKeysUn["octave"][sel].astype(np.float32)
    maxd = (d * np.float32(1.2) ** lvl * rng.uniform(0.95, 1.05, len(sel))).astype(np.float32)
    normal = (world / d[:, None] + rng.normal(0, 0.3, world.shape)).astype(np.float32)
    desc2 = last.mDescriptors[sel].copy()
    desc2[:, rng.integers(0, 32, 3)] ^= rng.integers(0, 256, (len(sel), 3)).astype(np.uint8) & 0x21
    two = lambda a: np.concatenate([a, a])
    mp = ola.MapPointGeom(two(world), two(normal), two(maxd), two(maxd) / np.float32(1.2) ** 7, np.concatenate([last.mDescriptors[sel], desc2]),
                          skip=rng.random(2 * len(sel)) < 0.1)
    R, t = _rot(0.002, -0.003, 0.001), np.array([0.01, -0.004, 0.02], np.float32)
    Scw = np.eye(4, dtype=np.float32); Scw[:3, :3] = np.float32(scale) * R; Scw[:3, 3] = np.float32(scale) * t
    mp.world = ((mp.world - t) @ R).astype(np.float32)
    matched0 = rng.random(kf.N) < 0.15                                       # vpMatched as SearchByBoW left it
    n_o, km_o, m_o = oracle.search_by_projection_sim3(kf, Scw, mp, matched0, th)
    matched = matched0.copy()
    n_g, km_g = ola.ORBmatcher(0.75, True).SearchByProjectionSim3(kf, Scw, mp, matched, th)
    assert n_g == n_o and np.array_equal(km_g, km_o) and np.array_equal(matched.astype(np.uint8), m_o)
    assert n_g > 100 and not (km_g[matched0] >= 0).any()
    first, second = km_g[(km_g >= 0) & (km_g < len(sel))], km_g[km_g >= len(sel)]
    assert len(second) > 10                                                  # second copies that found another key point: the greedy state matters


@pytest.mark.parametrize("th,s12", [(7.5, 1.0), (10.0, 0.98)])
def test_search_by_sim3(oracle, th, s12):
    """SearchBySim3 (loop closing): both key frames' map points projected into the other under the similarity, mutual agreement"""
    last, cur = _frames(oracle, seed=97)
    kf1, kf2 = _as_kf(last), _as_kf(cur)
    rng = np.random.default_rng(31)
    # KF2 = the frame shifted by 3 px; its map points are KF1's points displaced by that shift, attached to the features a projection
    # search pairs them with (any consistent attachment would do: the test needs two key frames observing one scene)
    _, m = oracle.search_by_projection(copy.deepcopy(cur), last, 7, False)
    has = m >= 0
    kf2.mp_valid = has.copy()
    kf2.mp_world = np.zeros((kf2.N, 3), np.float32)
    w = last.mp_world[m[has]].copy(); w[:, 0] += np.float32(3 * 0.54 / 718.856) * w[:, 2]
    kf2.mp_world[has] = w
    kf2.mp_desc = kf2.mDescriptors.copy()
    for k in (kf1, kf2):
        d = np.linalg.norm(np.where(k.mp_valid[:, None], k.mp_world, 1), axis=1).astype(np.float32)
        lv = k.mvKeysUn["octave"].astype(np.float32)
        k.mp_maxd = (d * np.float32(1.2) ** lv * rng.uniform(0.95, 1.05, k.N)).astype(np.float32)
        k.mp_mind = (k.mp_maxd / np.float32(1.2) ** 7).astype(np.float32)
        k.mp_bad = rng.random(k.N) < 0.05
        k.mTcw = np.eye(4, dtype=np.float32)
    kf2.mTcw[0, 3] = 0.004                                   # the two cameras see the world from slightly different poses
    R12 = _rot(0.001, 0.002, -0.001)
    t12 = np.array([-0.002, 0.001, 0.003], np.float32)
    matches12 = np.full(kf1.N, -1, np.int64)                 # some features are matched already (to a feature of KF2, or to a point KF2 does not see)
    pre = np.flatnonzero(rng.random(kf1.N) < 0.05)
    vals = rng.integers(0, kf2.N + 20, len(pre)); vals[vals >= kf2.N] = -2
    matches12[pre] = vals
    n_o, v1_o, v2_o, m_o = oracle.search_by_sim3(kf1, kf2, matches12, s12, R12, t12, th)
    mg = matches12.copy()
    n_g, v1_g, v2_g = ola.ORBmatcher(0.75, True).SearchBySim3(kf1, kf2, mg, s12, R12, t12, th)
    assert n_g == n_o and np.array_equal(v1_g, v1_o) and np.array_equal(v2_g, v2_o) and np.array_equal(mg, m_o)
    assert (v1_g >= 0).sum() > 50 and (v2_g >= 0).sum() > 50


def test_reference_signature_calls(tmp_path):
    """The reference's own call signatures (include/orbline_reference_api.hpp) on a device: Frame / KeyFrame / MapPoint stand-ins through
    SearchByProjection x2, SearchByBoW, match x3 and StereoFrameFeatures; every projected point must come back on its own key point."""
    import os, subprocess, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host_cpu import _build_reference_api
    out = subprocess.run([_build_reference_api(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert out.returncode == 0 and b"REFERENCE_API_OK" in out.stdout, out.stdout
