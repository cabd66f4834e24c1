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
    c.mp_maxd, c.mp_mind = p(getattr(v, "mp_maxd", None), np.float32), p(getattr(v, "mp_mind", None), np.float32)
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


def _search_by_projection_match12(self, CurrentFrame, LastFrame, th, bMono, match12):
    """int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono, map<int,int>& match12),
    src/ORBmatcher.cc:1474-1618 (Tracking::TrackWithMotionModelWithLine, src/Tracking.cc:1296,1302) -> olf_search_by_projection_match12.
    `match12` (a dict) is cleared and filled like the reference's map: key = CurrentFrame feature, value = the FIRST LastFrame feature it was
    matched with; keys in ascending order, as a std::map iterates.  Returns (nmatches, matches) like the four-argument overload."""
    keep = []
    cur, last = _view_c(CurrentFrame, keep), _view_c(LastFrame, keep)
    matches, m12, n = np.full(CurrentFrame.N, -1, np.int32), np.full(CurrentFrame.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_by_projection_match12(_ctx(self._context).handle, cur, last, float(th), int(bool(bMono)), int(bool(self.mbCheckOrientation)),
                                                 ptr(matches), ptr(m12), ptr(n)), "olf_search_by_projection_match12")
    match12.clear()
    for i2 in np.nonzero(m12 >= 0)[0]:
        match12[int(i2)] = int(m12[i2])
    return int(n[0]), matches


def _search_for_initialization(self, F1, F2, vbPrevMatched, windowSize=10):
    """int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12,
    int windowSize), src/ORBmatcher.cc:407-522 (monocular initialisation) -> olf_search_for_initialization.  vbPrevMatched: float32 [N1, 2],
    updated in place like the reference's vector.  Returns (nmatches, vnMatches12)."""
    keep = []
    f1, f2 = _view_c(F1, keep), _view_c(F2, keep)
    if not (isinstance(vbPrevMatched, np.ndarray) and vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags.c_contiguous
            and vbPrevMatched.shape == (F1.N, 2)):
        raise ValueError("vbPrevMatched: contiguous float32 array of shape (F1.N, 2)")
    m, n = np.full(F1.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_for_initialization(_ctx(self._context).handle, f1, f2, ptr(vbPrevMatched), int(windowSize), float(self.mfNNratio),
                                              int(bool(self.mbCheckOrientation)), ptr(m), ptr(n)), "olf_search_for_initialization")
    return int(n[0]), m


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


def isInFrustum(F, vpMapPoints, viewingCosLimit=0.5):
    """bool Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit), src/Frame.cc:388-444, for every point of a MapPointGeom (world, normal,
    maxd = mfMaxDistance, mind = mfMinDistance, descriptor) -> olf_is_in_frustum (host arithmetic in the library).  Returns the MapPointView
    that SearchByProjection(Frame, vector<MapPoint*>) reads: mbTrackInView, mnTrackScaleLevel, mTrackViewCos, mTrackProjX / Y / XR."""
    mp = vpMapPoints
    keep = []
    f = _view_c(F, keep)
    a = np.ascontiguousarray
    inview, level = np.zeros(mp.n, np.uint8), np.zeros(mp.n, np.int32)
    cosv, proj3 = np.zeros(mp.n, np.float32), np.zeros((mp.n, 3), np.float32)
    arrs = [a(mp.world, np.float32), a(mp.normal, np.float32), a(mp.maxd, np.float32), a(mp.mind, np.float32)]
    check(lib().olf_is_in_frustum(f, mp.n, *(ptr(x) for x in arrs), float(viewingCosLimit), ptr(inview), ptr(level), ptr(cosv), ptr(proj3)),
          "olf_is_in_frustum")
    return MapPointView(mp.descriptor, proj3[:, 0].copy(), proj3[:, 1].copy(), proj3[:, 2].copy(), level, cosv, mbTrackInView=inview.astype(bool),
                        isBad=mp.skip)


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


def _search_by_projection_kf(self, CurrentFrame, pKF, sAlreadyFound, th, ORBdist):
    """int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th,
    const int ORBdist), src/ORBmatcher.cc:1620-1747 (relocalisation) -> olf_search_by_projection_kf.  pKF: KeyFrameView with mp_valid /
    mp_bad / mp_world / mp_desc and mp_maxd / mp_mind (mfMaxDistance / mfMinDistance); sAlreadyFound: bool mask over pKF's features.
    Returns (nmatches, matches) with matches[i2] = pKF feature index; CurrentFrame.mp_valid is updated like mvpMapPoints."""
    keep = []
    cur, kf = _view_c(CurrentFrame, keep), _view_c(pKF, keep)
    found = None if sAlreadyFound is None else np.ascontiguousarray(sAlreadyFound, np.uint8)
    matches, n = np.full(CurrentFrame.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_by_projection_kf(_ctx(self._context).handle, cur, kf, None if found is None else ptr(found), float(th), int(ORBdist),
                                            int(bool(self.mbCheckOrientation)), ptr(matches), ptr(n)), "olf_search_by_projection_kf")
    return int(n[0]), matches


def _search_by_projection_dispatch(self, a, b, *args):
    """The reference overloads SearchByProjection on the second argument: a Frame (:1330; with a trailing map<int,int>& match12 :1474), a
    vector<MapPoint*> (:47) or a KeyFrame* (:1620; with a cv::Mat Scw first :292 -- see SearchByProjectionSim3)."""
    if isinstance(b, KeyFrameView):
        return _search_by_projection_kf(self, a, b, *args)
    return _search_by_projection_dispatch2(self, a, b, *args)


def _search_by_projection_dispatch2(self, a, b, th=1.0, bMono=False, match12=None):
    if isinstance(b, MapPointView):
        return _search_local_map(self, a, b, th)
    if match12 is not None:
        return _search_by_projection_match12(self, a, b, th, bMono, match12)
    return _search_by_projection(self, a, b, th, bMono)


ORBmatcher.SearchByProjection = _search_by_projection_dispatch
def _search_by_bow_kf(self, pKF1, pKF2):
    """int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), src/ORBmatcher.cc:524-657 (loop closing)
    -> olf_search_by_bow_kf.  Returns (nmatches, vpMatches12) with vpMatches12[idx1] = feature index idx2 of pKF2 whose map point is taken
    (-1 = NULL)."""
    keep = []
    k1, k2 = _view_c(pKF1, keep), _view_c(pKF2, keep)
    m12, n = np.full(pKF1.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_by_bow_kf(_ctx(self._context).handle, k1, k2, float(self.mfNNratio), int(bool(self.mbCheckOrientation)), ptr(m12), ptr(n)),
          "olf_search_by_bow_kf")
    return int(n[0]), m12


def _search_by_bow_dispatch(self, pKF, other):
    return _search_by_bow_kf(self, pKF, other) if isinstance(other, KeyFrameView) else _search_by_bow(self, pKF, other)


def _search_for_triangulation(self, pKF1, pKF2, F12, bOnlyStereo, Cw=None):
    """int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs,
    const bool bOnlyStereo), src/ORBmatcher.cc:659-825 (LocalMapping::CreateNewMapPoints) -> olf_search_for_triangulation.  Cw =
    pKF1->GetCameraCenter() (default: from pKF1.mTcw); pKF2.mTcw gives R2w / t2w.  Returns (nmatches, vMatchedPairs) with vMatchedPairs =
    [(idx1, idx2), ...] in idx1 order."""
    keep = []
    k1, k2 = _view_c(pKF1, keep), _view_c(pKF2, keep)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    cw = None if Cw is None else np.ascontiguousarray(Cw, np.float32).reshape(3)
    m12, n = np.full(pKF1.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_for_triangulation(_ctx(self._context).handle, k1, k2, ptr(F), None if cw is None else ptr(cw), int(bool(bOnlyStereo)),
                                             int(bool(self.mbCheckOrientation)), ptr(m12), ptr(n)), "olf_search_for_triangulation")
    return int(n[0]), [(int(i), int(m12[i])) for i in range(pKF1.N) if m12[i] >= 0]


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
    """The search part of int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th), src/ORBmatcher.cc:827-948
    -> olf_fuse_search: per map point the most similar key point of pKF inside the projection window.  Returns (bestIdx, bestDist) arrays
    (-1 / 256 where a gate rejects the point).  The reference's loop then fuses when bestDist <= TH_LOW (:950-972: Replace / AddObservation
    on the map, host code); that mutation never feeds back into another point's search, so the loop body splits exactly there."""
    mp = vpMapPoints
    keep = []
    kf = _view_c(pKF, keep)
    a = np.ascontiguousarray
    arrs = [a(mp.skip, np.uint8), a(mp.world, np.float32), a(mp.normal, np.float32), a(mp.maxd, np.float32), a(mp.mind, np.float32),
            a(mp.descriptor, np.uint8)]
    ow = None if Ow is None else a(Ow, np.float32).reshape(3)
    bi, bd = np.full(mp.n, -1, np.int32), np.full(mp.n, 256, np.int32)
    check(lib().olf_fuse_search(_ctx(self._context).handle, kf, mp.n, *(ptr(x) for x in arrs), float(th), None if ow is None else ptr(ow), ptr(bi),
                                ptr(bd)), "olf_fuse_search")
    return bi, bd


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


INT_MAX = 2147483647


def _fuse_search_sim3(self, pKF, Scw, vpPoints, th):
    """The search part of int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th,
    vector<MapPoint*> &vpReplacePoint), src/ORBmatcher.cc:977-1102 (loop correction) -> olf_fuse_search_sim3: per point the most similar key
    point of pKF inside the projection window under the Sim3 pose, (bestIdx, bestDist), (-1, INT_MAX) where a gate rejects the point.
    vpPoints.skip = isBad() || spAlreadyFound.count(pMP).  The reference then records a replacement / adds the observation when
    bestDist <= TH_LOW (:1086-1099, host code on the map)."""
    mp = vpPoints
    keep = []
    kf = _view_c(pKF, keep)
    a = np.ascontiguousarray
    S = a(Scw, np.float32).reshape(16)
    arrs = [a(mp.skip, np.uint8), a(mp.world, np.float32), a(mp.normal, np.float32), a(mp.maxd, np.float32), a(mp.mind, np.float32),
            a(mp.descriptor, np.uint8)]
    bi, bd = np.full(mp.n, -1, np.int32), np.full(mp.n, INT_MAX, np.int32)
    check(lib().olf_fuse_search_sim3(_ctx(self._context).handle, kf, ptr(S), mp.n, *(ptr(x) for x in arrs), float(th), ptr(bi), ptr(bd)),
          "olf_fuse_search_sim3")
    return bi, bd.astype(np.int64)


def _search_by_projection_sim3(self, pKF, Scw, vpPoints, vpMatched, th):
    """int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th),
    src/ORBmatcher.cc:292-405 (LoopClosing::ComputeSim3, src/LoopClosing.cc:381) -> olf_search_by_projection_sim3.  vpPoints: MapPointView with
    skip = isBad() || (the point is already in vpMatched); vpMatched: bool array over pKF's key points (vpMatched[idx] != NULL), updated in
    place.  Returns (nmatches, matches) with matches[idx] = index into vpPoints that key point idx received (-1 = none)."""
    mp = vpPoints
    keep = []
    kf = _view_c(pKF, keep)
    a = np.ascontiguousarray
    S = a(Scw, np.float32).reshape(16)
    arrs = [a(mp.skip, np.uint8), a(mp.world, np.float32), a(mp.normal, np.float32), a(mp.maxd, np.float32), a(mp.mind, np.float32),
            a(mp.descriptor, np.uint8)]
    if not (isinstance(vpMatched, np.ndarray) and vpMatched.dtype == np.bool_ and vpMatched.flags.c_contiguous and len(vpMatched) == pKF.N):
        raise ValueError("vpMatched: a C-contiguous bool array with one entry per key point of pKF (updated in place)")
    matches, n = np.full(pKF.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_by_projection_sim3(_ctx(self._context).handle, kf, ptr(S), mp.n, *(ptr(x) for x in arrs), float(th), vpMatched.ctypes.data,
                                              ptr(matches), ptr(n)), "olf_search_by_projection_sim3")
    return int(n[0]), matches


ORBmatcher.SearchByProjectionSim3 = _search_by_projection_sim3


def _search_by_sim3(self, pKF1, pKF2, vpMatches12, s12, R12, t12, th):
    """int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
    const cv::Mat &t12, const float th), src/ORBmatcher.cc:1104-1328 -> olf_search_by_sim3.  pKF1 / pKF2: KeyFrameView (mp_valid, mp_bad,
    mp_world, mp_desc, mp_maxd, mp_mind, mTcw).  vpMatches12: int array over pKF1's features, the index in pKF2 of the feature whose map
    point is already matched (pMP->GetIndexInKeyFrame(pKF2)), -1 for none, or -2 for "matched to a map point that pKF2 does not observe";
    updated in place like the reference's vector.  Returns (nFound, vnMatch1, vnMatch2)."""
    keep = []
    k1, k2 = _view_c(pKF1, keep), _view_c(pKF2, keep)
    m12 = np.ascontiguousarray(vpMatches12, np.int32)
    R, t = np.ascontiguousarray(R12, np.float32).reshape(9), np.ascontiguousarray(t12, np.float32).reshape(3)
    v1, v2, n = np.full(pKF1.N, -1, np.int32), np.full(pKF2.N, -1, np.int32), np.zeros(1, np.int32)
    check(lib().olf_search_by_sim3(_ctx(self._context).handle, k1, k2, ptr(m12), float(s12), ptr(R), ptr(t), float(th), ptr(v1), ptr(v2), ptr(n)),
          "olf_search_by_sim3")
    vpMatches12[...] = m12
    return int(n[0]), v1, v2


ORBmatcher.SearchForTriangulation = _search_for_triangulation
ORBmatcher.SearchForInitialization = _search_for_initialization
ORBmatcher.FuseSearchSim3 = _fuse_search_sim3
ORBmatcher.SearchBySim3 = _search_by_sim3
ORBmatcher.FuseSearch = _fuse_search


def match_maplines(maplines_desc, frame_desc_l, nnr, context=None):
    """int match(const std::vector<MapLine*>&, Frame&, float nnr, std::vector<int>& matches_12), src/LineMatcher.cpp:64-73: the local map
    lines' descriptors against mDescriptors_Line with matchNNR (the code after the early return there is dead)."""
    return matchNNR(maplines_desc, frame_desc_l, nnr, context)


ORBmatcher.SearchByBoW = _search_by_bow_dispatch
