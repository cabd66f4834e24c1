"""ctypes binding of oracle/liboracle.so -- the CPU oracle (test infrastructure only)."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, ROOT)
from orb_line_slam_amd._lib import KEYPOINT_DTYPE, KEYLINE_DTYPE, OrbParams, LineParams, StereoParams, OlfParams  # record layouts only

_L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
_L.orc_fast_atan2.restype = C.c_float
_L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def orb_params(nfeatures=2000, scale=1.2, nlevels=8, ini=20, mn=7):
    return OrbParams(nfeatures, scale, nlevels, ini, mn)


def orb_tables(p):
    n = p.nlevels
    sf, inv, s2, is2 = [np.zeros(n, np.float32) for _ in range(4)]
    npl = np.zeros(n, np.int32)
    umax = np.zeros(16, np.int32)
    _L.orc_orb_tables(C.byref(p), _p(sf), _p(inv), _p(s2), _p(is2), _p(npl), _p(umax))
    return sf, inv, s2, is2, npl, umax


def orb_level_sizes(p, w, h):
    lw = np.zeros(p.nlevels, np.int32)
    lh = np.zeros(p.nlevels, np.int32)
    _L.orc_orb_level_sizes(C.byref(p), w, h, _p(lw), _p(lh))
    return lw, lh


def orb_extract(img, p, debug=False, cap=None):
    """returns dict(kps, desc, [pyramid, blurred, candidates])"""
    img = np.ascontiguousarray(img)
    h, w = img.shape
    cap = cap or (p.nfeatures + 64)
    kps = np.zeros(cap, KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int()
    out = {}
    if debug:
        lw, lh = orb_level_sizes(p, w, h)
        tot = int((lw.astype(np.int64) * lh).sum())
        pyr = np.zeros(tot, np.uint8)
        blur = np.zeros(tot, np.uint8)
        ccap = 65536
        cand = np.zeros((p.nlevels, ccap, 3), np.int32)
        cc = np.zeros(p.nlevels, np.int32)
        rc = _L.orc_orb_extract(_p(img), w, h, w, C.byref(p), _p(kps), _p(desc), cap, C.byref(n), _p(pyr), _p(blur), _p(cand), _p(cc), ccap)
        off = 0
        out["pyramid"], out["blurred"], out["candidates"] = [], [], []
        for l in range(p.nlevels):
            sz = int(lw[l]) * int(lh[l])
            out["pyramid"].append(pyr[off:off + sz].reshape(lh[l], lw[l]))
            out["blurred"].append(blur[off:off + sz].reshape(lh[l], lw[l]))
            out["candidates"].append(cand[l, :cc[l]].copy())
            off += sz
    else:
        rc = _L.orc_orb_extract(_p(img), w, h, w, C.byref(p), _p(kps), _p(desc), cap, C.byref(n), None, None, None, None, 0)
    assert rc == 0, rc
    out["kps"] = kps[:n.value].copy()
    out["desc"] = desc[:n.value].copy()
    return out


def hamming256(a, b):
    return int(_L.orc_hamming256(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b))))


def resize_linear(src, dw, dh, scale_x=None, scale_y=None):
    src = np.ascontiguousarray(src)
    sh, sw = src.shape
    if scale_x is None:
        scale_x = 1.0 / (dw / sw)
    if scale_y is None:
        scale_y = 1.0 / (dh / sh)
    dst = np.zeros((dh, dw), np.uint8)
    _L.orc_resize_linear(_p(src), sw, sh, _p(dst), dw, dh, C.c_double(scale_x), C.c_double(scale_y))
    return dst


def gaussian_blur(src, ksize, sigma):
    src = np.ascontiguousarray(src)
    h, w = src.shape
    dst = np.zeros((h, w), np.uint8)
    taps = np.zeros(ksize, np.int32)
    _L.orc_gaussian_blur(_p(src), w, h, _p(dst), ksize, C.c_double(sigma), _p(taps))
    return dst, taps


def fast_atan2(y, x):
    return float(_L.orc_fast_atan2(y, x))


def match_bf(d1, d2, nnr, best_lr=True):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    m12 = np.full(len(d1), -1, np.int32)
    _L.orc_match_bf(_p(d1), len(d1), _p(d2), len(d2), C.c_float(nnr), int(best_lr), _p(m12))
    return m12


def knn2(d1, d2):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    n = len(d1)
    i0, a, b = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    _L.orc_knn2(_p(d1), n, _p(d2), len(d2), _p(i0), _p(a), _p(b))
    return i0, a, b


def full_params(nfeatures=2000, nlines=500, fx=718.856, bf=386.1448):
    from orb_line_slam_amd._lib import default_params
    try:
        p = default_params()
    except Exception:
        raise
    p.orb.nfeatures = nfeatures
    p.line.lsd_nfeatures = nlines
    p.stereo.fx, p.stereo.bf = fx, bf
    return p


def stereo_points(imgL, imgR, p, cap=None):
    imgL = np.ascontiguousarray(imgL); imgR = np.ascontiguousarray(imgR)
    h, w = imgL.shape
    cap = cap or p.orb.nfeatures + 64
    kl, kr = np.zeros(cap, KEYPOINT_DTYPE), np.zeros(cap, KEYPOINT_DTYPE)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    nl, nr = C.c_int(), C.c_int()
    ur, dp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    sad = np.zeros(cap, np.int32)
    rc = _L.orc_stereo_points(_p(imgL), _p(imgR), w, h, C.byref(p), _p(kl), _p(dl), C.byref(nl), _p(kr), _p(dr), C.byref(nr), cap,
                              _p(ur), _p(dp), _p(sad))
    assert rc == 0, rc
    n, m = nl.value, nr.value
    return dict(kpsL=kl[:n], descL=dl[:n], kpsR=kr[:m], descR=dr[:m], uRight=ur[:n], depth=dp[:n], sad=sad[:n])
