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


class ORBmatcher:
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30   # src/ORBmatcher.cc:39-41

    def __init__(self, nnratio=0.6, checkOri=True, context=None):
        self.mfNNratio, self.mbCheckOrientation, self._context = float(nnratio), bool(checkOri), context

    @staticmethod
    def DescriptorDistance(a, b):
        return distance(a, b)
