"""Host-side mirrors of the reference's descriptor matchers on top of the C ABI.

  LineMatcher free functions  include/LineMatcher.h:57-69, src/LineMatcher.cpp:42-150
  ORBmatcher                  include/ORBmatcher.h:37-103 (constants + DescriptorDistance; the searches that
                              need MapPoint state keep their greedy resolution on the host)
All distances are computed on the GPU (csrc/match.hip).
"""
import numpy as np
from . import _lib
from ._lib import check, lib, ptr

_shared_ctx = None


def _ctx(context=None):
    """matchers are stateless in the reference; they borrow a small context for stream + scratch"""
    global _shared_ctx
    if context is not None:
        return context
    if _shared_ctx is None:
        _shared_ctx = _lib.Context(_lib.default_params(), 640, 480, 1)
    return _shared_ctx


def _desc(d):
    d = np.ascontiguousarray(d, dtype=np.uint8)
    if d.ndim != 2 or (d.shape[0] and d.shape[1] != 32):
        raise ValueError("descriptors must be an (N, 32) uint8 array (cv::Mat N x 32 CV_8U)")
    return d


def knn2(desc1, desc2, context=None):
    """cv::BFMatcher(NORM_HAMMING).knnMatch(desc1, desc2, k=2): (idx0, dist0, dist1) per query row."""
    d1, d2 = _desc(desc1), _desc(desc2)
    n = d1.shape[0]
    idx0, dist0, dist1 = (np.full(n, -1, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32))
    check(lib().olf_knn2(_ctx(context).handle, ptr(d1), n, ptr(d2), d2.shape[0], ptr(idx0), ptr(dist0), ptr(dist1)), "olf_knn2")
    return idx0, dist0, dist1


def matchNNR(desc1, desc2, nnr, context=None):
    """src/LineMatcher.cpp:42-62 -> (n_matches, matches_12)"""
    return match(desc1, desc2, nnr, best_lr_matches=False, context=context)


def match(desc1, desc2, nnr, best_lr_matches=True, context=None):
    """match(desc1, desc2, nnr, matches_12), src/LineMatcher.cpp:104-132 -> (n_matches, matches_12).
    best_lr_matches is Config::bestLRMatches() (default true, src/Config.cpp:47)."""
    d1, d2 = _desc(desc1), _desc(desc2)
    m12 = np.full(d1.shape[0], -1, np.int32)
    check(lib().olf_match_bf(_ctx(context).handle, ptr(d1), d1.shape[0], ptr(d2), d2.shape[0], float(nnr), int(bool(best_lr_matches)),
                             ptr(m12)), "olf_match_bf")
    return int((m12 >= 0).sum()), m12


def distance_matrix(desc1, desc2, context=None):
    """ORB_SLAM2::distance / ORBmatcher::DescriptorDistance over all pairs -> (N1, N2) uint16"""
    d1, d2 = _desc(desc1), _desc(desc2)
    out = np.zeros((d1.shape[0], d2.shape[0]), np.uint16)
    check(lib().olf_hamming_matrix(_ctx(context).handle, ptr(d1), d1.shape[0], ptr(d2), d2.shape[0], ptr(out)), "olf_hamming_matrix")
    return out


def distance(a, b, context=None):
    """int distance(const cv::Mat&, const cv::Mat&), src/LineMatcher.cpp:134-150"""
    return int(distance_matrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32), context)[0, 0])


def ComputeDistinctiveDescriptors(observations, context=None):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:254-318) / MapLine::ComputeDistinctiveDescriptors (src/MapLine.cc:257-322)
    for a batch of landmarks.  observations: list of (N_p, 32) uint8 arrays (the descriptors observing landmark p, bad key frames already
    dropped).  Returns BestIdx per landmark (-1 where the list is empty: the reference returns early and keeps the old descriptor)."""
    obs = [np.ascontiguousarray(o, np.uint8).reshape(-1, 32) for o in observations]
    offs = np.zeros(len(obs) + 1, np.int32)
    offs[1:] = np.cumsum([len(o) for o in obs])
    desc = np.ascontiguousarray(np.concatenate(obs)) if len(obs) and offs[-1] else np.zeros((0, 32), np.uint8)
    best = np.full(len(obs), -1, np.int32)
    if len(obs):
        check(lib().olf_distinctive_descriptors(_ctx(context).handle, ptr(desc), ptr(offs), len(obs), ptr(best)), "olf_distinctive_descriptors")
    return best


class ORBmatcher:
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30   # src/ORBmatcher.cc:39-41

    def __init__(self, nnratio=0.6, checkOri=True, context=None):
        self.mfNNratio, self.mbCheckOrientation, self._context = float(nnratio), bool(checkOri), context

    @staticmethod
    def DescriptorDistance(a, b):
        return distance(a, b)


# ------------------------------------------------------------------------------------------------------------------
# ORBmatcher searches that need Frame / MapPoint state.  The reference walks C++ objects (Frame, KeyFrame, MapPoint*);
# here a frame is a FrameView: plain arrays with the members those functions read.  Host code builds the candidate
# lists (Frame::GetFeaturesInArea / the BoW feature vectors) and replays the greedy, order-dependent resolution exactly
# as the reference does; every descriptor distance comes from the GPU (olf_match_candidates).
class FrameView:
    """The Frame / KeyFrame members read by SearchByProjection (src/ORBmatcher.cc:1330-1472) and SearchByBoW (:161-290).

    mvKeysUn / mvKeys : KEYPOINT_DTYPE arrays;  mDescriptors : (N, 32) uint8;  mvuRight : (N,) float32
    mp_valid[i]  : mvpMapPoints[i] != NULL        mp_world[i] : pMP->GetWorldPos() (float32 x3)
    mp_desc[i]   : pMP->GetDescriptor()           mp_obs[i]   : pMP->Observations() > 0      mp_bad[i] : pMP->isBad()
    mvbOutlier   : (N,) bool                      mTcw : (4, 4) float32
    fx, fy, cx, cy, mbf, mb, mnMinX, mnMaxX, mnMinY, mnMaxY : camera / image bounds;  mvScaleFactors : (nlevels,) float32
    mFeatVec     : {node id: [feature indices]} (DBoW2::FeatureVector), only for SearchByBoW
    """
    FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48

    def __init__(self, mvKeysUn, mDescriptors, mvuRight=None, mvScaleFactors=None, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157,
                 mbf=386.1448, bounds=(0.0, 1241.0, 0.0, 376.0), mTcw=None, mFeatVec=None):
        self.mvKeysUn = np.ascontiguousarray(mvKeysUn)
        self.mvKeys = self.mvKeysUn          # identical when the camera has no distortion (src/Frame.cc:601-605)
        self.N = len(self.mvKeysUn)
        self.mDescriptors = np.ascontiguousarray(mDescriptors, np.uint8).reshape(self.N, 32)
        self.mvuRight = np.full(self.N, -1.0, np.float32) if mvuRight is None else np.ascontiguousarray(mvuRight, np.float32)
        self.mvScaleFactors = np.ascontiguousarray(mvScaleFactors if mvScaleFactors is not None else 1.2 ** np.arange(8), np.float32)
        f32 = np.float32
        self.fx, self.fy, self.cx, self.cy, self.mbf = f32(fx), f32(fy), f32(cx), f32(cy), f32(mbf)
        self.mb = f32(self.mbf / self.fx)
        self.mnMinX, self.mnMaxX, self.mnMinY, self.mnMaxY = (f32(v) for v in bounds)
        self.mfGridElementWidthInv = f32(f32(self.FRAME_GRID_COLS) / f32(self.mnMaxX - self.mnMinX))
        self.mfGridElementHeightInv = f32(f32(self.FRAME_GRID_ROWS) / f32(self.mnMaxY - self.mnMinY))
        self.mTcw = np.eye(4, dtype=np.float32) if mTcw is None else np.ascontiguousarray(mTcw, np.float32)
        self.mp_valid = np.zeros(self.N, bool)
        self.mp_world = np.zeros((self.N, 3), np.float32)
        self.mp_desc = np.zeros((self.N, 32), np.uint8)
        self.mp_obs = np.zeros(self.N, bool)
        self.mp_bad = np.zeros(self.N, bool)
        self.mvbOutlier = np.zeros(self.N, bool)
        self.mFeatVec = mFeatVec or {}
        self.AssignFeaturesToGrid()

    def AssignFeaturesToGrid(self):
        """src/Frame.cc:334-349 + PosInGrid :572-582 (C round(): half away from zero)"""
        self.mGrid = [[[] for _ in range(self.FRAME_GRID_ROWS)] for _ in range(self.FRAME_GRID_COLS)]
        f32 = np.float32
        for i in range(self.N):
            px = float(f32(f32(self.mvKeysUn["x"][i] - self.mnMinX) * self.mfGridElementWidthInv))
            py = float(f32(f32(self.mvKeysUn["y"][i] - self.mnMinY) * self.mfGridElementHeightInv))
            gx, gy = int(np.floor(abs(px) + 0.5) * np.sign(px)), int(np.floor(abs(py) + 0.5) * np.sign(py))
            if 0 <= gx < self.FRAME_GRID_COLS and 0 <= gy < self.FRAME_GRID_ROWS:
                self.mGrid[gx][gy].append(i)

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1):
        """src/Frame.cc:517-570"""
        f32 = np.float32
        x, y, r = f32(x), f32(y), f32(r)
        C, R = self.FRAME_GRID_COLS, self.FRAME_GRID_ROWS
        nMinCellX = max(0, int(np.floor(f32(f32(f32(x - self.mnMinX) - r) * self.mfGridElementWidthInv))))
        if nMinCellX >= C:
            return []
        nMaxCellX = min(C - 1, int(np.ceil(f32(f32(f32(x - self.mnMinX) + r) * self.mfGridElementWidthInv))))
        if nMaxCellX < 0:
            return []
        nMinCellY = max(0, int(np.floor(f32(f32(f32(y - self.mnMinY) - r) * self.mfGridElementHeightInv))))
        if nMinCellY >= R:
            return []
        nMaxCellY = min(R - 1, int(np.ceil(f32(f32(f32(y - self.mnMinY) + r) * self.mfGridElementHeightInv))))
        if nMaxCellY < 0:
            return []
        bCheckLevels = (minLevel > 0) or (maxLevel >= 0)
        out = []
        kx, ky, ko = self.mvKeysUn["x"], self.mvKeysUn["y"], self.mvKeysUn["octave"]
        for ix in range(nMinCellX, nMaxCellX + 1):
            for iy in range(nMinCellY, nMaxCellY + 1):
                for j in self.mGrid[ix][iy]:
                    if bCheckLevels:
                        if ko[j] < minLevel:
                            continue
                        if maxLevel >= 0 and ko[j] > maxLevel:
                            continue
                    if abs(f32(kx[j] - x)) < r and abs(f32(ky[j] - y)) < r:
                        out.append(j)
        return out


class KeyFrameView(FrameView):
    """A FrameView that stands for a KeyFrame* argument (the reference overloads SearchByBoW / SearchByProjection on Frame& / KeyFrame*).
    mp_maxd / mp_mind : (N,) float32, the map points' mfMaxDistance / mfMinDistance (MapPoint::UpdateNormalAndDepth)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.mp_maxd = np.zeros(self.N, np.float32)
        self.mp_mind = np.zeros(self.N, np.float32)


def _candidate_distances(descQ, lists, descT, context=None):
    """GPU Hamming distances for CSR candidate lists; returns a list of uint16 arrays (one per query)."""
    offs = np.zeros(len(lists) + 1, np.int32)
    offs[1:] = np.cumsum([len(l) for l in lists])
    cand = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32) for l in lists]) if offs[-1] else np.zeros(0, np.int32), np.int32)
    dist = np.zeros(int(offs[-1]), np.uint16)
    dq, dt = _desc(descQ), _desc(descT)
    if len(lists) and offs[-1]:
        check(lib().olf_match_candidates(_ctx(context).handle, ptr(dq), dq.shape[0], ptr(dt), dt.shape[0], ptr(offs), ptr(cand), ptr(dist)),
              "olf_match_candidates")
    return [dist[offs[i]:offs[i + 1]] for i in range(len(lists))]


def _c_round(v):
    v = float(v)
    return int(np.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def ComputeThreeMaxima(histo):
    """src/ORBmatcher.cc:1749-1790"""
    max1 = max2 = max3 = 0
    ind1 = ind2 = ind3 = -1
    for i, h in enumerate(histo):
        s = len(h)
        if s > max1:
            max3, max2, max1 = max2, max1, s
            ind3, ind2, ind1 = ind2, ind1, i
        elif s > max2:
            max3, max2 = max2, s
            ind3, ind2 = ind2, i
        elif s > max3:
            max3, ind3 = s, i
    if max2 < np.float32(0.1) * np.float32(max1):
        ind2 = ind3 = -1
    elif max3 < np.float32(0.1) * np.float32(max1):
        ind3 = -1
    return ind1, ind2, ind3


def _view_c(v, keep):
    """olf_frame_view of a FrameView / KeyFrameView; every array it points to is appended to `keep` (alive until the call returns).  The bool
    state arrays are passed as they are (one byte per element), so the library's updates land in the view's own arrays."""
    from ._lib import FrameViewC
    c = FrameViewC()

    def p(a, dt=None):
        if a is None:
            return None
        b = np.ascontiguousarray(a if dt is None else np.asarray(a, dt))
        keep.append(b)
        return b.ctypes.data

    def state(name):
        a = getattr(v, name, None)
        if a is None:
            return None
        if not (isinstance(a, np.ndarray) and a.dtype == np.bool_ and a.flags.c_contiguous):
            a = np.ascontiguousarray(a, bool)
            setattr(v, name, a)
        keep.append(a)
        return a.ctypes.data
    c.keys, c.desc, c.uright, c.n = p(v.mvKeysUn), p(v.mDescriptors), p(v.mvuRight, np.float32), v.N
    c.mp_valid, c.mp_obs, c.mp_bad, c.outlier = state("mp_valid"), state("mp_obs"), state("mp_bad"), state("mvbOutlier")
    c.mp_world, c.mp_desc, c.Tcw = p(v.mp_world, np.float32), p(v.mp_desc, np.uint8), p(v.mTcw, np.float32)
    c.fx, c.fy, c.cx, c.cy, c.mbf = float(v.fx), float(v.fy), float(v.cx), float(v.cy), float(v.mbf)
    c.minX, c.maxX, c.minY, c.maxY = float(v.mnMinX), float(v.mnMaxX), float(v.mnMinY), float(v.mnMaxY)
    c.scale_factors, c.n_levels = p(v.mvScaleFactors, np.float32), len(v.mvScaleFactors)
    nodes = sorted(v.mFeatVec)
    offs = np.zeros(len(nodes) + 1, np.int32)
    offs[1:] = np.cumsum([len(v.mFeatVec[k]) for k in nodes])
    feats = np.array([i for k in nodes for i in v.mFeatVec[k]], np.int32)
    c.fv_nodes, c.fv_offsets, c.fv_features, c.fv_n = p(np.array(nodes, np.int32)), p(offs), p(feats), len(nodes)
    return c


def _search_by_projection(self, CurrentFrame, LastFrame, th, bMono):
    """int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono),
    src/ORBmatcher.cc:1330-1472 -> olf_search_by_projection (host candidate lists and resolution in csrc/search_host.cpp, distances on the
    GPU).  Returns (nmatches, matches) with matches[i2] = index i of the LastFrame map point assigned to CurrentFrame feature i2 (-1 = none);
    CurrentFrame.mp_valid / mp_obs are updated like mvpMapPoints."""
    keep = []
    cur, last = _view_c(CurrentFrame, keep), _view_c(LastFrame, keep)
    matches, n = np.full(CurrentFrame.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_by_projection(_ctx(self._context).handle, cur, last, float(th), int(bool(bMono)), int(bool(self.mbCheckOrientation)),
                                         ptr(matches), ptr(n)), "olf_search_by_projection")
    return int(n[0]), matches


def _search_by_bow(self, pKF, F):
    """int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), src/ORBmatcher.cc:161-290 ->
    olf_search_by_bow.  Returns (nmatches, vpMapPointMatches) with vpMapPointMatches[iF] = KF feature index whose map point was matched
    (-1 = none)."""
    keep = []
    kf, f = _view_c(pKF, keep), _view_c(F, keep)
    matched, n = np.full(F.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_by_bow(_ctx(self._context).handle, kf, f, float(self.mfNNratio), int(bool(self.mbCheckOrientation)), ptr(matched),
                                  ptr(n)), "olf_search_by_bow")
    return int(n[0]), matched


class MapPointView:
    """The MapPoint members read by SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:47-131), gathered by the
    host (mutex-guarded in the reference, src/MapPoint.cc:321-325) into SoA buffers:
    mbTrackInView, isBad : (n,) bool;  mnTrackScaleLevel : (n,) int32;  mTrackViewCos, mTrackProjX, mTrackProjY, mTrackProjXR : (n,) float32;
    descriptor : (n, 32) uint8 (GetDescriptor());  obs : (n,) bool (Observations() > 0)."""

    def __init__(self, descriptor, mTrackProjX, mTrackProjY, mTrackProjXR, mnTrackScaleLevel, mTrackViewCos, mbTrackInView=None, isBad=None,
                 obs=None):
        self.descriptor = np.ascontiguousarray(descriptor, np.uint8).reshape(-1, 32)
        self.n = n = len(self.descriptor)
        f = lambda v: np.ascontiguousarray(v, np.float32).reshape(n)
        self.mTrackProjX, self.mTrackProjY, self.mTrackProjXR, self.mTrackViewCos = f(mTrackProjX), f(mTrackProjY), f(mTrackProjXR), f(mTrackViewCos)
        self.mnTrackScaleLevel = np.ascontiguousarray(mnTrackScaleLevel, np.int32).reshape(n)
        b = lambda v, d: np.full(n, d, bool) if v is None else np.ascontiguousarray(v, bool).reshape(n)
        self.mbTrackInView, self.isBad, self.obs = b(mbTrackInView, True), b(isBad, False), b(obs, True)


def RadiusByViewingCos(viewCos):
    """src/ORBmatcher.cc:133-139 (the float is compared with the double literal 0.998)"""
    return np.float32(2.5) if float(np.float32(viewCos)) > 0.998 else np.float32(4.0)


def _search_local_map(self, F, vpMapPoints, th=1.0):
    """int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th), src/ORBmatcher.cc:47-131
    (Tracking::SearchLocalPoints, every frame) -> olf_search_local_map.  Returns (nmatches, matches) with matches[idx] = index into
    vpMapPoints assigned to feature idx (-1 = none); F.mp_valid / F.mp_obs are updated like F.mvpMapPoints."""
    mp = vpMapPoints
    keep = []
    f = _view_c(F, keep)
    a = np.ascontiguousarray
    proj3 = a(np.stack([mp.mTrackProjX, mp.mTrackProjY, mp.mTrackProjXR], 1), np.float32)
    arrs = [a(mp.mbTrackInView, np.uint8), a(mp.isBad, np.uint8), a(mp.mnTrackScaleLevel, np.int32), a(mp.mTrackViewCos, np.float32), proj3,
            a(mp.descriptor, np.uint8), a(mp.obs, np.uint8)]
    matches, n = np.full(F.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_local_map(_ctx(self._context).handle, f, mp.n, *(ptr(x) for x in arrs), float(th), float(self.mfNNratio), ptr(matches),
                                     ptr(n)), "olf_search_local_map")
    return int(n[0]), matches


_libm = None


def _logf(x):
    """glibc logf (MapPoint::PredictScale calls std::log on a float, src/MapPoint.cc:422): numpy's float32 log is not guaranteed to be it"""
    global _libm
    if _libm is None:
        import ctypes
        _libm = ctypes.CDLL("libm.so.6")
        _libm.logf.restype = ctypes.c_float
        _libm.logf.argtypes = [ctypes.c_float]
    return np.float32(_libm.logf(float(np.float32(x))))


def _search_by_projection_kf(self, CurrentFrame, pKF, sAlreadyFound, th, ORBdist):
    """int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th,
    const int ORBdist), src/ORBmatcher.cc:1620-1747 (relocalisation).  pKF: KeyFrameView with mp_valid / mp_bad / mp_world / mp_desc and
    mp_maxd / mp_mind (mfMaxDistance / mfMinDistance); sAlreadyFound: bool mask over pKF's features.  Returns (nmatches, matches) with
    matches[i2] = pKF feature index; CurrentFrame.mp_valid is updated like mvpMapPoints."""
    f32 = np.float32
    HL = self.HISTO_LENGTH
    rotHist = [[] for _ in range(HL)]
    factor = f32(1.0) / f32(HL)
    Rcw, tcw = CurrentFrame.mTcw[:3, :3], CurrentFrame.mTcw[:3, 3]
    Ow = (-(Rcw.T.astype(np.float64) @ tcw.astype(np.float64))).astype(f32)
    nLevels = len(CurrentFrame.mvScaleFactors)
    logScale = _logf(CurrentFrame.mvScaleFactors[1]) if nLevels > 1 else f32(1.0)       # mfLogScaleFactor = log(mfScaleFactor), src/Frame.cc:157
    found = np.zeros(pKF.N, bool) if sAlreadyFound is None else np.asarray(sAlreadyFound, bool)
    queries, lists = [], []
    for i in range(pKF.N):
        if not pKF.mp_valid[i] or pKF.mp_bad[i] or found[i]:
            continue
        x3Dw = pKF.mp_world[i]
        x3Dc = (Rcw.astype(np.float64) @ x3Dw.astype(np.float64) + tcw.astype(np.float64)).astype(f32)
        xc, yc = x3Dc[0], x3Dc[1]
        invzc = f32(1.0 / np.float64(x3Dc[2])) if x3Dc[2] != 0 else f32(np.inf)
        u = f32(f32(f32(CurrentFrame.fx * xc) * invzc) + CurrentFrame.cx)
        v = f32(f32(f32(CurrentFrame.fy * yc) * invzc) + CurrentFrame.cy)
        if u < CurrentFrame.mnMinX or u > CurrentFrame.mnMaxX or v < CurrentFrame.mnMinY or v > CurrentFrame.mnMaxY:
            continue
        PO = (x3Dw - Ow).astype(f32)
        dist3D = f32(np.sqrt(np.sum(PO.astype(np.float64) ** 2)))
        maxDistance, minDistance = f32(f32(1.2) * pKF.mp_maxd[i]), f32(f32(0.8) * pKF.mp_mind[i])
        if dist3D < minDistance or dist3D > maxDistance:
            continue
        ratio = f32(pKF.mp_maxd[i] / dist3D)
        nPredictedLevel = int(np.ceil(f32(_logf(ratio) / logScale)))
        nPredictedLevel = 0 if nPredictedLevel < 0 else min(nPredictedLevel, nLevels - 1)
        radius = f32(f32(th) * CurrentFrame.mvScaleFactors[nPredictedLevel])
        idx = CurrentFrame.GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1)
        if not idx:
            continue
        queries.append(i); lists.append(idx)
    dists = _candidate_distances(pKF.mp_desc[queries] if queries else np.zeros((0, 32), np.uint8), lists, CurrentFrame.mDescriptors, self._context)
    matches = np.full(CurrentFrame.N, -1, np.int32)
    nmatches = 0
    for qi, i in enumerate(queries):
        bestDist, bestIdx2 = 256, -1
        for i2, dist in zip(lists[qi], dists[qi]):
            if CurrentFrame.mp_valid[i2]:
                continue
            if int(dist) < bestDist:
                bestDist, bestIdx2 = int(dist), i2
        if bestDist <= ORBdist:
            CurrentFrame.mp_valid[bestIdx2] = True
            matches[bestIdx2] = i
            nmatches += 1
            if self.mbCheckOrientation:
                rot = f32(pKF.mvKeysUn["angle"][i] - CurrentFrame.mvKeysUn["angle"][bestIdx2])
                if rot < 0.0:
                    rot = f32(rot + f32(360.0))
                b = _c_round(f32(rot * factor))
                if b == HL:
                    b = 0
                rotHist[b].append(bestIdx2)
    if self.mbCheckOrientation:
        ind = ComputeThreeMaxima(rotHist)
        for b in range(HL):
            if b not in ind:
                for j in rotHist[b]:
                    CurrentFrame.mp_valid[j] = False
                    matches[j] = -1
                    nmatches -= 1
    return nmatches, matches


def _search_by_projection_dispatch(self, a, b, *args):
    """The reference overloads SearchByProjection on the second argument: a Frame (:1330), a vector<MapPoint*> (:47) or a KeyFrame* (:1620)."""
    if isinstance(b, KeyFrameView):
        return _search_by_projection_kf(self, a, b, *args)
    return _search_by_projection_dispatch2(self, a, b, *args)


def _search_by_projection_dispatch2(self, a, b, th=1.0, bMono=False):
    if isinstance(b, MapPointView):
        return _search_local_map(self, a, b, th)
    return _search_by_projection(self, a, b, th, bMono)


ORBmatcher.SearchByProjection = _search_by_projection_dispatch
def _search_by_bow_kf(self, pKF1, pKF2):
    """int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), src/ORBmatcher.cc:524-657 (loop closing).
    Returns (nmatches, vpMatches12) with vpMatches12[idx1] = feature index idx2 of pKF2 whose map point is taken (-1 = NULL)."""
    f32 = np.float32
    HL = self.HISTO_LENGTH
    rotHist = [[] for _ in range(HL)]
    factor = f32(1.0) / f32(HL)
    common = sorted(set(pKF1.mFeatVec) & set(pKF2.mFeatVec))
    queries, lists = [], []
    for node in common:
        for idx1 in pKF1.mFeatVec[node]:
            if not pKF1.mp_valid[idx1] or pKF1.mp_bad[idx1]:
                continue
            queries.append(idx1); lists.append(list(pKF2.mFeatVec[node]))
    dists = _candidate_distances(pKF1.mDescriptors[queries] if queries else np.zeros((0, 32), np.uint8), lists, pKF2.mDescriptors, self._context)
    matches12 = np.full(pKF1.N, -1, np.int32)
    vbMatched2 = np.zeros(pKF2.N, bool)
    nmatches = 0
    for qi, idx1 in enumerate(queries):
        bestDist1, bestIdx2, bestDist2 = 256, -1, 256
        for idx2, dist in zip(lists[qi], dists[qi]):
            if vbMatched2[idx2] or not pKF2.mp_valid[idx2] or pKF2.mp_bad[idx2]:
                continue
            dist = int(dist)
            if dist < bestDist1:
                bestDist2, bestDist1, bestIdx2 = bestDist1, dist, idx2
            elif dist < bestDist2:
                bestDist2 = dist
        if bestDist1 < self.TH_LOW and f32(bestDist1) < f32(self.mfNNratio) * f32(bestDist2):
            matches12[idx1] = bestIdx2
            vbMatched2[bestIdx2] = True
            if self.mbCheckOrientation:
                rot = f32(pKF1.mvKeysUn["angle"][idx1] - pKF2.mvKeysUn["angle"][bestIdx2])
                if rot < 0.0:
                    rot = f32(rot + f32(360.0))
                b = _c_round(f32(rot * factor))
                if b == HL:
                    b = 0
                rotHist[b].append(idx1)
            nmatches += 1
    if self.mbCheckOrientation:
        ind = ComputeThreeMaxima(rotHist)
        for b in range(HL):
            if b in ind:
                continue
            for j in rotHist[b]:
                matches12[j] = -1
                nmatches -= 1
    return nmatches, matches12


def _search_by_bow_dispatch(self, pKF, other):
    return _search_by_bow_kf(self, pKF, other) if isinstance(other, KeyFrameView) else _search_by_bow(self, pKF, other)


def _epipolar_ok(kp1, kp2, F12, levelSigma2):
    """ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:142-161 (float arithmetic, the threshold product in double)"""
    f32 = np.float32
    x1, y1, x2, y2 = f32(kp1["x"]), f32(kp1["y"]), f32(kp2["x"]), f32(kp2["y"])
    a = f32(f32(f32(x1 * F12[0, 0]) + f32(y1 * F12[1, 0])) + F12[2, 0])
    b = f32(f32(f32(x1 * F12[0, 1]) + f32(y1 * F12[1, 1])) + F12[2, 1])
    c = f32(f32(f32(x1 * F12[0, 2]) + f32(y1 * F12[1, 2])) + F12[2, 2])
    num = f32(f32(f32(a * x2) + f32(b * y2)) + c)
    den = f32(f32(a * a) + f32(b * b))
    if den == 0:
        return False
    dsqr = f32(f32(num * num) / den)
    return float(dsqr) < 3.84 * float(levelSigma2[int(kp2["octave"])])


def _search_for_triangulation(self, pKF1, pKF2, F12, bOnlyStereo, Cw=None):
    """int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs,
    const bool bOnlyStereo), src/ORBmatcher.cc:659-825 (LocalMapping::CreateNewMapPoints).  Cw = pKF1->GetCameraCenter() (default: from
    pKF1.mTcw); pKF2.mTcw gives R2w / t2w.  Returns (nmatches, vMatchedPairs) with vMatchedPairs = [(idx1, idx2), ...] in idx1 order."""
    f32 = np.float32
    HL = self.HISTO_LENGTH
    F12 = np.ascontiguousarray(F12, f32).reshape(3, 3)
    if Cw is None:
        R1, t1 = pKF1.mTcw[:3, :3], pKF1.mTcw[:3, 3]
        Cw = (-(R1.T.astype(np.float64) @ t1.astype(np.float64))).astype(f32)
    R2w, t2w = pKF2.mTcw[:3, :3], pKF2.mTcw[:3, 3]
    C2 = (R2w.astype(np.float64) @ np.asarray(Cw, f32).astype(np.float64) + t2w.astype(np.float64)).astype(f32)
    invz = f32(f32(1.0) / C2[2])
    ex = f32(f32(f32(pKF2.fx * C2[0]) * invz) + pKF2.cx)
    ey = f32(f32(f32(pKF2.fy * C2[1]) * invz) + pKF2.cy)
    sigma2 = (pKF2.mvScaleFactors * pKF2.mvScaleFactors).astype(f32)          # mvLevelSigma2, src/ORBextractor.cc:430-436
    rotHist = [[] for _ in range(HL)]
    factor = f32(1.0) / f32(HL)
    common = sorted(set(pKF1.mFeatVec) & set(pKF2.mFeatVec))
    queries, lists = [], []
    for node in common:
        for idx1 in pKF1.mFeatVec[node]:
            if pKF1.mp_valid[idx1]:
                continue
            if bOnlyStereo and not pKF1.mvuRight[idx1] >= 0:
                continue
            queries.append(idx1); lists.append(list(pKF2.mFeatVec[node]))
    dists = _candidate_distances(pKF1.mDescriptors[queries] if queries else np.zeros((0, 32), np.uint8), lists, pKF2.mDescriptors, self._context)
    vMatches12 = np.full(pKF1.N, -1, np.int32)
    nmatches = 0
    for qi, idx1 in enumerate(queries):
        bStereo1 = bool(pKF1.mvuRight[idx1] >= 0)
        kp1 = pKF1.mvKeysUn[idx1]
        bestDist, bestIdx2 = self.TH_LOW, -1
        for idx2, dist in zip(lists[qi], dists[qi]):
            if pKF2.mp_valid[idx2]:                      # (vbMatched2 is never set in the reference)
                continue
            bStereo2 = bool(pKF2.mvuRight[idx2] >= 0)
            if bOnlyStereo and not bStereo2:
                continue
            dist = int(dist)
            if dist > self.TH_LOW or dist > bestDist:
                continue
            kp2 = pKF2.mvKeysUn[idx2]
            if not bStereo1 and not bStereo2:
                dx, dy = f32(ex - kp2["x"]), f32(ey - kp2["y"])
                if f32(f32(dx * dx) + f32(dy * dy)) < f32(f32(100) * pKF2.mvScaleFactors[int(kp2["octave"])]):
                    continue
            if _epipolar_ok(kp1, kp2, F12, sigma2):
                bestIdx2, bestDist = idx2, dist
        if bestIdx2 >= 0:
            vMatches12[idx1] = bestIdx2
            nmatches += 1
            if self.mbCheckOrientation:
                rot = f32(kp1["angle"] - pKF2.mvKeysUn["angle"][bestIdx2])
                if rot < 0.0:
                    rot = f32(rot + f32(360.0))
                b = _c_round(f32(rot * factor))
                if b == HL:
                    b = 0
                rotHist[b].append(idx1)
    if self.mbCheckOrientation:
        ind = ComputeThreeMaxima(rotHist)
        for b in range(HL):
            if b in ind:
                continue
            for j in rotHist[b]:
                vMatches12[j] = -1
                nmatches -= 1
    return nmatches, [(int(i), int(vMatches12[i])) for i in range(pKF1.N) if vMatches12[i] >= 0]


class MapPointGeom:
    """The MapPoint members read by the search part of Fuse (src/ORBmatcher.cc:827-948): skip = !pMP || isBad() || IsInKeyFrame(pKF),
    world = GetWorldPos(), normal = GetNormal(), maxd / mind = mfMaxDistance / mfMinDistance, descriptor = GetDescriptor()."""

    def __init__(self, world, normal, maxd, mind, descriptor, skip=None):
        self.descriptor = np.ascontiguousarray(descriptor, np.uint8).reshape(-1, 32)
        self.n = n = len(self.descriptor)
        self.world = np.ascontiguousarray(world, np.float32).reshape(n, 3)
        self.normal = np.ascontiguousarray(normal, np.float32).reshape(n, 3)
        self.maxd, self.mind = np.ascontiguousarray(maxd, np.float32).reshape(n), np.ascontiguousarray(mind, np.float32).reshape(n)
        self.skip = np.zeros(n, bool) if skip is None else np.ascontiguousarray(skip, bool).reshape(n)


def _fuse_search(self, pKF, vpMapPoints, th=3.0, Ow=None):
    """The search part of int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th), src/ORBmatcher.cc:827-948:
    per map point the most similar key point of pKF inside the projection window.  Returns (bestIdx, bestDist) arrays (-1 / 256 where a gate
    rejects the point).  The reference's loop then fuses when bestDist <= TH_LOW (:950-972: Replace / AddObservation on the map, host
    code); that mutation never feeds back into another point's search, so the loop body splits exactly there."""
    f32 = np.float32
    mp = vpMapPoints
    Rcw, tcw = pKF.mTcw[:3, :3], pKF.mTcw[:3, 3]
    if Ow is None:
        Ow = (-(Rcw.T.astype(np.float64) @ tcw.astype(np.float64))).astype(f32)
    Ow = np.asarray(Ow, f32)
    nLevels = len(pKF.mvScaleFactors)
    logScale = _logf(pKF.mvScaleFactors[1]) if nLevels > 1 else f32(1.0)
    sigma2 = (pKF.mvScaleFactors * pKF.mvScaleFactors).astype(f32)
    invSigma2 = (f32(1.0) / sigma2).astype(f32)                                   # mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]
    queries, lists, meta = [], [], []
    for i in range(mp.n):
        if mp.skip[i]:
            continue
        p3Dw = mp.world[i]
        p3Dc = (Rcw.astype(np.float64) @ p3Dw.astype(np.float64) + tcw.astype(np.float64)).astype(f32)
        if p3Dc[2] < 0.0:
            continue
        invz = f32(f32(1) / p3Dc[2]) if p3Dc[2] != 0 else f32(np.inf)
        x, y = f32(p3Dc[0] * invz), f32(p3Dc[1] * invz)
        u, v = f32(f32(pKF.fx * x) + pKF.cx), f32(f32(pKF.fy * y) + pKF.cy)
        if not (u >= pKF.mnMinX and u < pKF.mnMaxX and v >= pKF.mnMinY and v < pKF.mnMaxY):
            continue
        ur = f32(u - f32(pKF.mbf * invz))
        maxDistance, minDistance = f32(f32(1.2) * mp.maxd[i]), f32(f32(0.8) * mp.mind[i])
        PO = (p3Dw - Ow).astype(f32)
        dist3D = f32(np.sqrt(np.sum(PO.astype(np.float64) ** 2)))
        if dist3D < minDistance or dist3D > maxDistance:
            continue
        if float(np.sum(PO.astype(np.float64) * mp.normal[i].astype(np.float64))) < 0.5 * float(dist3D):
            continue
        ratio = f32(mp.maxd[i] / dist3D)
        lvl = int(np.ceil(f32(_logf(ratio) / logScale)))
        lvl = 0 if lvl < 0 else min(lvl, nLevels - 1)
        radius = f32(f32(th) * pKF.mvScaleFactors[lvl])
        idx = pKF.GetFeaturesInArea(u, v, radius)
        if not idx:
            continue
        queries.append(i); lists.append(idx); meta.append((u, v, ur, lvl))
    dists = _candidate_distances(mp.descriptor[queries] if queries else np.zeros((0, 32), np.uint8), lists, pKF.mDescriptors, self._context)
    bestIdx, bestDist = np.full(mp.n, -1, np.int32), np.full(mp.n, 256, np.int32)
    kx, ky, ko = pKF.mvKeysUn["x"], pKF.mvKeysUn["y"], pKF.mvKeysUn["octave"]
    for qi, i in enumerate(queries):
        u, v, ur, lvl = meta[qi]
        bd, bi = 256, -1
        for idx, dist in zip(lists[qi], dists[qi]):
            kpLevel = int(ko[idx])
            if kpLevel < lvl - 1 or kpLevel > lvl:
                continue
            ex, ey = f32(u - kx[idx]), f32(v - ky[idx])
            if pKF.mvuRight[idx] >= 0:
                er = f32(ur - pKF.mvuRight[idx])
                e2 = f32(f32(f32(ex * ex) + f32(ey * ey)) + f32(er * er))
                if float(f32(e2 * invSigma2[kpLevel])) > 7.8:
                    continue
            else:
                e2 = f32(f32(ex * ex) + f32(ey * ey))
                if float(f32(e2 * invSigma2[kpLevel])) > 5.99:
                    continue
            if int(dist) < bd:
                bd, bi = int(dist), idx
        bestIdx[i], bestDist[i] = bi, bd
    return bestIdx, bestDist


def _scale(m, s):
    """cv::Mat (CV_32F) times a scalar: MatExpr scaling, every element times the double factor rounded to float"""
    return (np.asarray(m, np.float32) * np.float32(s)).astype(np.float32)


def _gemv(R, v, t=None, alpha=1.0):
    """alpha * R * v (+ t) on CV_32F operands: double accumulation, one rounding (cv::gemm)"""
    acc = alpha * (np.asarray(R, np.float64) @ np.asarray(v, np.float64))
    if t is not None:
        acc = acc + np.asarray(t, np.float64)
    return acc.astype(np.float32)


def Sim3Decompose(Scw):
    """Scw -> (Rcw, tcw, Ow) as at the top of ORBmatcher::Fuse(pKF, Scw, ...), src/ORBmatcher.cc:985-989"""
    Scw = np.asarray(Scw, np.float32)
    sRcw = Scw[:3, :3]
    scw = np.float32(np.sqrt(np.sum(sRcw[0].astype(np.float64) ** 2)))
    inv = np.float32(1.0 / np.float64(scw))
    Rcw, tcw = _scale(sRcw, inv), _scale(Scw[:3, 3], inv)
    return Rcw, tcw, _gemv(Rcw.T, tcw, alpha=-1.0)


def _predict_scale(maxd, dist3D, logScale, nLevels):
    """MapPoint::PredictScale, src/MapPoint.cc:414-429"""
    lvl = int(np.ceil(np.float32(_logf(np.float32(maxd / dist3D)) / logScale)))
    return 0 if lvl < 0 else min(lvl, nLevels - 1)


def _project_in_image(KF, p3Dc):
    """pinhole projection + KeyFrame::IsInImage; (u, v) or None"""
    f32 = np.float32
    if p3Dc[2] < 0.0:
        return None
    invz = f32(1.0 / np.float64(p3Dc[2])) if p3Dc[2] != 0 else f32(np.inf)
    x, y = f32(p3Dc[0] * invz), f32(p3Dc[1] * invz)
    u, v = f32(f32(KF.fx * x) + KF.cx), f32(f32(KF.fy * y) + KF.cy)
    if not (u >= KF.mnMinX and u < KF.mnMaxX and v >= KF.mnMinY and v < KF.mnMaxY):
        return None
    return u, v


def _best_in_lists(KF, queries, lists, levels, dists, n, empty):
    """per query the candidate with octave in [level - 1, level] and the smallest distance (first wins)"""
    bestIdx, bestDist = np.full(n, -1, np.int32), np.full(n, empty, np.int64)
    ko = KF.mvKeysUn["octave"]
    for qi, i in enumerate(queries):
        bd, bi = empty, -1
        for idx, dist in zip(lists[qi], dists[qi]):
            if ko[idx] < levels[qi] - 1 or ko[idx] > levels[qi]:
                continue
            if int(dist) < bd:
                bd, bi = int(dist), idx
        bestIdx[i], bestDist[i] = bi, bd
    return bestIdx, bestDist


INT_MAX = 2147483647


def _fuse_search_sim3(self, pKF, Scw, vpPoints, th):
    """The search part of int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th,
    vector<MapPoint*> &vpReplacePoint), src/ORBmatcher.cc:977-1102 (loop correction): per point the most similar key point of pKF inside the
    projection window under the Sim3 pose, (bestIdx, bestDist), (-1, INT_MAX) where a gate rejects the point.  vpPoints.skip =
    isBad() || spAlreadyFound.count(pMP).  The reference then records a replacement / adds the observation when bestDist <= TH_LOW
    (:1086-1099, host code on the map)."""
    f32 = np.float32
    mp = vpPoints
    Rcw, tcw, Ow = Sim3Decompose(Scw)
    nLevels = len(pKF.mvScaleFactors)
    logScale = _logf(pKF.mvScaleFactors[1]) if nLevels > 1 else f32(1.0)
    queries, lists, levels = [], [], []
    for i in range(mp.n):
        if mp.skip[i]:
            continue
        p3Dw = mp.world[i]
        uv = _project_in_image(pKF, _gemv(Rcw, p3Dw, tcw))
        if uv is None:
            continue
        maxDistance, minDistance = f32(f32(1.2) * mp.maxd[i]), f32(f32(0.8) * mp.mind[i])
        PO = (p3Dw - Ow).astype(f32)
        dist3D = f32(np.sqrt(np.sum(PO.astype(np.float64) ** 2)))
        if dist3D < minDistance or dist3D > maxDistance:
            continue
        if float(np.sum(PO.astype(np.float64) * mp.normal[i].astype(np.float64))) < 0.5 * float(dist3D):
            continue
        lvl = _predict_scale(mp.maxd[i], dist3D, logScale, nLevels)
        idx = pKF.GetFeaturesInArea(uv[0], uv[1], f32(f32(th) * pKF.mvScaleFactors[lvl]))
        if not idx:
            continue
        queries.append(i); lists.append(idx); levels.append(lvl)
    dists = _candidate_distances(mp.descriptor[queries] if queries else np.zeros((0, 32), np.uint8), lists, pKF.mDescriptors, self._context)
    return _best_in_lists(pKF, queries, lists, levels, dists, mp.n, INT_MAX)


def _search_by_sim3(self, pKF1, pKF2, vpMatches12, s12, R12, t12, th):
    """int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
    const cv::Mat &t12, const float th), src/ORBmatcher.cc:1104-1328.  pKF1 / pKF2: KeyFrameView (mp_valid, mp_bad, mp_world, mp_desc, mp_maxd,
    mp_mind, mTcw).  vpMatches12: int array over pKF1's features, the index in pKF2 of the feature whose map point is already matched
    (pMP->GetIndexInKeyFrame(pKF2)), -1 for none, or -2 for "matched to a map point that pKF2 does not observe"; updated in place like the
    reference's vector.  Returns (nFound, vnMatch1, vnMatch2)."""
    f32 = np.float32
    N1, N2 = pKF1.N, pKF2.N
    m12 = np.asarray(vpMatches12)
    already1 = m12 != -1
    already2 = np.zeros(N2, bool)
    already2[m12[(m12 >= 0) & (m12 < N2)]] = True
    sR12 = _scale(R12, f32(s12))
    sR21 = _scale(np.asarray(R12, f32).T, f32(1.0 / np.float64(f32(s12))))
    t21 = _gemv(sR21, t12, alpha=-1.0)
    nLevels = len(pKF1.mvScaleFactors)
    logScale = _logf(pKF1.mvScaleFactors[1]) if nLevels > 1 else f32(1.0)
    vn = []
    for src, dst, already, sR, t in ((pKF1, pKF2, already1, sR21, t21), (pKF2, pKF1, already2, sR12, np.asarray(t12, f32).reshape(3))):
        Rw, tw = src.mTcw[:3, :3], src.mTcw[:3, 3]
        queries, lists, levels = [], [], []
        for i in range(src.N):
            if not src.mp_valid[i] or already[i] or src.mp_bad[i]:
                continue
            pb = _gemv(sR, _gemv(Rw, src.mp_world[i], tw), t)
            uv = _project_in_image(dst, pb)
            if uv is None:
                continue
            maxDistance, minDistance = f32(f32(1.2) * src.mp_maxd[i]), f32(f32(0.8) * src.mp_mind[i])
            dist3D = f32(np.sqrt(np.sum(pb.astype(np.float64) ** 2)))
            if dist3D < minDistance or dist3D > maxDistance:
                continue
            lvl = _predict_scale(src.mp_maxd[i], dist3D, logScale, nLevels)
            idx = dst.GetFeaturesInArea(uv[0], uv[1], f32(f32(th) * dst.mvScaleFactors[lvl]))
            if not idx:
                continue
            queries.append(i); lists.append(idx); levels.append(lvl)
        dists = _candidate_distances(src.mp_desc[queries] if queries else np.zeros((0, 32), np.uint8), lists, dst.mDescriptors, self._context)
        bi, bd = _best_in_lists(dst, queries, lists, levels, dists, src.N, INT_MAX)
        vn.append(np.where(bd <= self.TH_HIGH, bi, -1).astype(np.int32))
    vnMatch1, vnMatch2 = vn
    nFound = 0
    for i1 in range(N1):
        idx2 = vnMatch1[i1]
        if idx2 >= 0 and vnMatch2[idx2] == i1:
            vpMatches12[i1] = idx2
            nFound += 1
    return nFound, vnMatch1, vnMatch2


ORBmatcher.SearchForTriangulation = _search_for_triangulation
ORBmatcher.FuseSearchSim3 = _fuse_search_sim3
ORBmatcher.SearchBySim3 = _search_by_sim3
ORBmatcher.FuseSearch = _fuse_search


def match_maplines(maplines_desc, frame_desc_l, nnr, context=None):
    """int match(const std::vector<MapLine*>&, Frame&, float nnr, std::vector<int>& matches_12), src/LineMatcher.cpp:64-73: the local map
    lines' descriptors against mDescriptors_Line with matchNNR (the code after the early return there is dead)."""
    return matchNNR(maplines_desc, frame_desc_l, nnr, context)


ORBmatcher.SearchByBoW = _search_by_bow_dispatch
