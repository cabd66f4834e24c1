"""Input conditioning ahead of the feature path, on the GPU.

  cvtColor(im, gray, CV_RGB2GRAY | CV_BGR2GRAY | CV_RGBA2GRAY | CV_BGRA2GRAY)   Tracking::GrabImageStereo, src/Tracking.cc:193-218
  remap(im, rect, M1, M2, INTER_LINEAR)                                         Examples/PL/PL_stereo_euroc.cc:136-137
"""
import numpy as np
from . import _lib
from ._lib import check, lib, ptr

RGB2GRAY, BGR2GRAY, RGBA2GRAY, BGRA2GRAY = 0, 1, 2, 3


def _ctx(context, w, h):
    return context if context is not None else _lib.Context(_lib.default_params(), w, h, 1)


def cvtColor(images, code, context=None):
    """images: (n, H, W, 3|4) uint8 -> (n, H, W) uint8.  Mirrors mbRGB handling: code RGB2GRAY when Camera.RGB = 1, else BGR2GRAY."""
    images = np.ascontiguousarray(images)
    n, h, w, cn = images.shape
    if images.dtype != np.uint8 or cn != (4 if code >= 2 else 3):
        raise TypeError("cvtColor: (n, H, W, 3|4) uint8 expected for this code")
    c = _ctx(context, w, h)
    if (c.width, c.height) != (w, h):
        raise ValueError("context size differs from the image size")
    out = np.zeros((n, h, w), np.uint8)
    check(lib().olf_cvt_gray(c.handle, ptr(images), int(code), n, ptr(out)), "olf_cvt_gray")
    return out


def remap(images, map1, map2, context=None):
    """images: (n, H, W) uint8; map1/map2: (H2, W2) float32 (CV_32FC1 x and y maps) -> (n, H2, W2) uint8, INTER_LINEAR, BORDER_CONSTANT 0."""
    images = np.ascontiguousarray(images)
    map1, map2 = np.ascontiguousarray(map1, np.float32), np.ascontiguousarray(map2, np.float32)
    n, h, w = images.shape
    dh, dw = map1.shape
    if images.dtype != np.uint8 or map2.shape != map1.shape:
        raise TypeError("remap: (n, H, W) uint8 images and two equal-shape float32 maps expected")
    c = _ctx(context, max(w, 640), max(h, 480))
    out = np.zeros((n, dh, dw), np.uint8)
    check(lib().olf_remap_linear(c.handle, ptr(images), w, h, ptr(map1), ptr(map2), dw, dh, n, ptr(out)), "olf_remap_linear")
    return out


def initUndistortRectifyMap(K, D, R, P, size, context=None):
    """cv::initUndistortRectifyMap(K, D, R, P[:3, :3], size, CV_32F) -> (map1, map2) float32 of shape (height, width); size = (width, height)
    like cv::Size.  Examples/PL/PL_stereo_euroc.cc:97-98."""
    import ctypes as C
    w, h = int(size[0]), int(size[1])
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3); R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    P = np.ascontiguousarray(np.asarray(P, np.float64)[:3, :3])
    D = np.ascontiguousarray(np.asarray(D, np.float64).reshape(-1))
    c = _ctx(context, max(w, 640), max(h, 480))
    m1, m2 = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    L = lib()
    L.olf_init_undistort_rectify_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    check(L.olf_init_undistort_rectify_map(c.handle, ptr(K), ptr(D), len(D), ptr(R), ptr(P), w, h, ptr(m1), ptr(m2)), "olf_init_undistort_rectify_map")
    return m1, m2
