"""Batched stereo front-end: the feature part of Frame::Frame (reference src/Frame.cc:136-221)
for many independent stereo pairs at once -- ExtractORB x2, ExtractLine x2, ComputeStereoMatches,
ComputeStereoMatches_Lines -- on one GPU.  Mirrors the members the reference's Tracking reads:
mvKeys / mvKeysRight / mDescriptors / mDescriptorsRight / mvuRight / mvDepth (+ the line members).
"""
import numpy as np
from . import _lib
from ._lib import KEYPOINT_DTYPE, DESC_BYTES, check, lib, ptr


class StereoFrames:
    """Results of one batch; arrays are (n_pairs, capacity[, ...]) with per-pair counts."""

    def __init__(self):
        self.N = self.Nr = None
        self.mvKeys = self.mvKeysRight = self.mDescriptors = self.mDescriptorsRight = None
        self.mvuRight = self.mvDepth = None

    def pair(self, i):
        n, nr = int(self.N[i]), int(self.Nr[i])
        return dict(mvKeys=self.mvKeys[i, :n], mDescriptors=self.mDescriptors[i, :n], mvKeysRight=self.mvKeysRight[i, :nr],
                    mDescriptorsRight=self.mDescriptorsRight[i, :nr], mvuRight=self.mvuRight[i, :n], mvDepth=self.mvDepth[i, :n])


class StereoFrontEnd:
    def __init__(self, params=None, width=1242, height=375, max_pairs=1):
        self.params = params or _lib.default_params()
        self.ctx = _lib.Context(self.params, width, height, 2 * max_pairs)
        self.width, self.height, self.max_pairs = width, height, max_pairs

    def stereo_points(self, images):
        """images: (2*n_pairs, H, W) uint8, image 2p = left, 2p+1 = right (host memory)."""
        images = np.ascontiguousarray(images)
        if images.dtype != np.uint8 or images.ndim != 3 or images.shape[0] % 2:
            raise TypeError("stereo_points: (2*n_pairs, H, W) uint8 expected")
        if images.shape[1:] != (self.height, self.width):
            # Frame::Frame throws when the sizes differ (src/Frame.cc:145-146)
            raise RuntimeError("[StereoFrame] Left and right images have different sizes")
        n_pairs = images.shape[0] // 2
        cap = self.ctx.orb_capacity
        kps = np.zeros((2 * n_pairs, cap), KEYPOINT_DTYPE)
        desc = np.zeros((2 * n_pairs, cap, DESC_BYTES), np.uint8)
        counts = np.zeros(2 * n_pairs, np.int32)
        ur = np.zeros((n_pairs, cap), np.float32)
        dp = np.zeros((n_pairs, cap), np.float32)
        check(lib().olf_stereo_points(self.ctx.handle, ptr(images), n_pairs, ptr(kps), ptr(desc), ptr(counts), ptr(ur), ptr(dp)),
              "olf_stereo_points")
        f = StereoFrames()
        f.N, f.Nr = counts[0::2].copy(), counts[1::2].copy()
        f.mvKeys, f.mvKeysRight = kps[0::2], kps[1::2]
        f.mDescriptors, f.mDescriptorsRight = desc[0::2], desc[1::2]
        f.mvuRight, f.mvDepth = ur, dp
        return f
