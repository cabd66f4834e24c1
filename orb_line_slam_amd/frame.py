"""Batched stereo front-end: the feature part of Frame::Frame (reference src/Frame.cc:136-221)
for many independent stereo pairs at once -- ExtractORB x2, ExtractLine x2, ComputeStereoMatches,
ComputeStereoMatches_Lines -- on one GPU.  Mirrors the members the reference's Tracking reads:
mvKeys / mvKeysRight / mDescriptors / mDescriptorsRight / mvuRight / mvDepth (+ the line members).
"""
import numpy as np
from . import _lib
import ctypes as C
from ._lib import KEYPOINT_DTYPE, KEYLINE_DTYPE, DESC_BYTES, FrameBuffers, check, lib, ptr


class StereoFrames:
    """Results of one batch; arrays are (n_pairs, capacity[, ...]) with per-pair counts."""

    def __init__(self):
        self.N = self.Nr = None
        self.mvKeys = self.mvKeysRight = self.mDescriptors = self.mDescriptorsRight = None
        self.mvuRight = self.mvDepth = None
        self.N_l = self.Nr_l = None
        self.mvKeys_Line = self.mvKeysRight_Line = self.mDescriptors_Line = self.mDescriptorsRight_Line = None
        self.line_matches_12 = self.mvDisparity_l = self.mvle_l = None

    def pair(self, i):
        n, nr = int(self.N[i]), int(self.Nr[i])
        d = self._pair_points(i, n, nr)
        if self.N_l is not None:
            nl, nrl = int(self.N_l[i]), int(self.Nr_l[i])
            d.update(mvKeys_Line=self.mvKeys_Line[i, :nl], mDescriptors_Line=self.mDescriptors_Line[i, :nl],
                     mvKeysRight_Line=self.mvKeysRight_Line[i, :nrl], mDescriptorsRight_Line=self.mDescriptorsRight_Line[i, :nrl],
                     line_matches_12=self.line_matches_12[i, :nl], mvDisparity_l=self.mvDisparity_l[i, :nl], mvle_l=self.mvle_l[i, :nl])
        return d

    def _pair_points(self, i, n, nr):
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

    def stereo_lines(self, kls, ldesc, lcounts):
        """ComputeStereoMatches_Lines on already extracted key lines: kls (2*n_pairs, cap), ldesc, lcounts."""
        kls, ldesc = np.ascontiguousarray(kls), np.ascontiguousarray(ldesc)
        lcounts = np.ascontiguousarray(lcounts, np.int32)
        n_pairs = kls.shape[0] // 2
        cap = self.ctx.line_capacity
        assert kls.shape[1] == cap
        m12 = np.full((n_pairs, cap), -1, np.int32)
        disp = np.zeros((n_pairs, cap, 2), np.float32)
        le = np.zeros((n_pairs, cap, 3), np.float64)
        check(lib().olf_stereo_lines(self.ctx.handle, n_pairs, ptr(kls), ptr(ldesc), ptr(lcounts), ptr(m12), ptr(disp), ptr(le)), "olf_stereo_lines")
        return m12, disp, le

    def frames(self, images):
        """The whole feature part of Frame::Frame (src/Frame.cc:136-221) for (2*n_pairs, H, W) uint8 host images."""
        images = np.ascontiguousarray(images)
        if images.dtype != np.uint8 or images.ndim != 3 or images.shape[0] % 2:
            raise TypeError("frames: (2*n_pairs, H, W) uint8 expected")
        if images.shape[1:] != (self.height, self.width):
            raise RuntimeError("[StereoFrame] Left and right images have different sizes")
        n_pairs = images.shape[0] // 2
        cap, lcap = self.ctx.orb_capacity, self.ctx.line_capacity
        kps = np.zeros((2 * n_pairs, cap), KEYPOINT_DTYPE)
        desc = np.zeros((2 * n_pairs, cap, DESC_BYTES), np.uint8)
        counts = np.zeros(2 * n_pairs, np.int32)
        ur, dp = np.zeros((n_pairs, cap), np.float32), np.zeros((n_pairs, cap), np.float32)
        kls = np.zeros((2 * n_pairs, lcap), KEYLINE_DTYPE)
        ldesc = np.zeros((2 * n_pairs, lcap, DESC_BYTES), np.uint8)
        lcounts = np.zeros(2 * n_pairs, np.int32)
        lm = np.full((n_pairs, lcap), -1, np.int32)
        ldisp = np.zeros((n_pairs, lcap, 2), np.float32)
        lle = np.zeros((n_pairs, lcap, 3), np.float64)
        fb = FrameBuffers(*[a.ctypes.data for a in (kps, desc, counts, ur, dp, kls, ldesc, lcounts, lm, ldisp, lle)])
        check(lib().olf_stereo_frames(self.ctx.handle, ptr(images), n_pairs, C.byref(fb)), "olf_stereo_frames")
        f = StereoFrames()
        f.N, f.Nr = counts[0::2].copy(), counts[1::2].copy()
        f.mvKeys, f.mvKeysRight = kps[0::2], kps[1::2]
        f.mDescriptors, f.mDescriptorsRight = desc[0::2], desc[1::2]
        f.mvuRight, f.mvDepth = ur, dp
        f.N_l, f.Nr_l = lcounts[0::2].copy(), lcounts[1::2].copy()
        f.mvKeys_Line, f.mvKeysRight_Line = kls[0::2], kls[1::2]
        f.mDescriptors_Line, f.mDescriptorsRight_Line = ldesc[0::2], ldesc[1::2]
        f.line_matches_12, f.mvDisparity_l, f.mvle_l = lm, ldisp, lle
        return f
