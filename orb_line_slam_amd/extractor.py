"""Host-side mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:52-118).

Same constructor arguments, same call shape (image, mask ignored -> key points + 32-byte
descriptors), same getters; `mvImagePyramid` is the public member Frame::ComputeStereoMatches reads
(src/Frame.cc:799-816), served lazily from the device copy.  All arithmetic happens in
liborbline_hip.so on the GPU.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import KEYPOINT_DTYPE, KEYLINE_DTYPE, DESC_BYTES, check, lib, ptr


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1   # include/ORBextractor.h:56 (unused by the reference as well)

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_images=2, context=None):
        p = _lib.default_params()
        p.orb.nfeatures, p.orb.scale_factor, p.orb.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        p.orb.ini_th_fast, p.orb.min_th_fast = int(iniThFAST), int(minThFAST)
        self._params, self._max_images, self._ctx = p, int(max_images), context
        self.nfeatures, self.scaleFactor, self.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        self.iniThFAST, self.minThFAST = int(iniThFAST), int(minThFAST)
        self._tables = None

    # -- context handling --------------------------------------------------------------------
    def _context(self, width, height, n_images):
        c = self._ctx
        if c is None or c.width != width or c.height != height or c.max_images < n_images:
            if c is not None:
                c.close()
            c = _lib.Context(self._params, width, height, max(n_images, self._max_images))
            self._ctx, self._tables = c, None
        return c

    def _get_tables(self):
        if self._tables is None:
            if self._ctx is None:
                raise RuntimeError("ORBextractor tables need a context: call the extractor once or pass context=")
            n = self.nlevels
            t = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32)]
            check(lib().olf_orb_scale_tables(self._ctx.handle, *[ptr(a) for a in t]), "olf_orb_scale_tables")
            self._tables = t
        return self._tables

    # -- reference getters (include/ORBextractor.h:68-91) ------------------------------------
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self._get_tables()[0].copy()

    def GetInverseScaleFactors(self):
        return self._get_tables()[1].copy()

    def GetScaleSigmaSquares(self):
        return self._get_tables()[2].copy()

    def GetInverseScaleSigmaSquares(self):
        return self._get_tables()[3].copy()

    def features_per_level(self):
        return self._get_tables()[4].copy()

    # -- operator() ---------------------------------------------------------------------------
    def __call__(self, image, mask=None):
        """image: (H, W) uint8.  Returns (keypoints[KEYPOINT_DTYPE], descriptors (N, 32) uint8).
        An empty image returns empty outputs (src/ORBextractor.cc:1048-1049); a non-8UC1 image is
        the reference's assert (:1052) -> TypeError here.  `mask` is ignored, as in the reference."""
        kps, desc, counts = self.extract_batch(np.asarray(image)[None] if np.asarray(image).size else np.zeros((0, 0, 0), np.uint8))
        if len(counts) == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, DESC_BYTES), np.uint8)
        n = int(counts[0])
        return kps[0, :n].copy(), desc[0, :n].copy()

    def extract_batch(self, images):
        """images: (n, H, W) uint8 -> (kps (n, cap), desc (n, cap, 32), counts (n,))."""
        images = np.asarray(images)
        if images.size == 0:
            return np.zeros((0, 0), KEYPOINT_DTYPE), np.zeros((0, 0, DESC_BYTES), np.uint8), np.zeros(0, np.int32)
        if images.dtype != np.uint8 or images.ndim != 3:
            raise TypeError("ORBextractor: image must be 8-bit single channel (CV_8UC1)")
        images = np.ascontiguousarray(images)
        n, h, w = images.shape
        ctx = self._context(w, h, n)
        cap = ctx.orb_capacity
        kps = np.zeros((n, cap), KEYPOINT_DTYPE)
        desc = np.zeros((n, cap, DESC_BYTES), np.uint8)
        counts = np.zeros(n, np.int32)
        check(lib().olf_orb_extract(ctx.handle, ptr(images), n, ptr(kps), ptr(desc), ptr(counts)), "olf_orb_extract")
        return kps, desc, counts

    # -- mvImagePyramid -----------------------------------------------------------------------
    def level_sizes(self):
        w = np.zeros(self.nlevels, np.int32)
        h = np.zeros(self.nlevels, np.int32)
        check(lib().olf_orb_level_sizes(self._ctx.handle, ptr(w), ptr(h)), "olf_orb_level_sizes")
        return w, h

    def pyramid_level(self, level, image=0, blurred=False):
        w, h = self.level_sizes()
        out = np.zeros((int(h[level]), int(w[level])), np.uint8)
        check(lib().olf_orb_pyramid_level(self._ctx.handle, image, level, int(blurred), ptr(out)), "olf_orb_pyramid_level")
        return out

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    def debug_candidates(self, level, image=0, cap=65536):
        xys = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        check(lib().olf_orb_debug_candidates(self._ctx.handle, image, level, ptr(xys), cap, C.byref(n)), "olf_orb_debug_candidates")
        return xys[:min(n.value, cap)]


class Lineextractor:
    """Mirror of ORB_SLAM2::Lineextractor (reference include/LineExtractor.h:40-72): LSD detection, top-N by
    response, LBD description.  Both reference constructors are supported: (nfeatures, length_th[, bFLD]) and the
    11-argument LSD form.  bFLD=True returns nothing, exactly like the reference (src/LineExtractor.cc:68)."""

    def __init__(self, lsd_nfeatures, llength_th, lsd_refine=0, lsd_scale=1.2, lsd_sigma_scale=0.6, lsd_quant=2.0, lsd_ang_th=22.5,
                 lsd_log_eps=1.0, lsd_density_th=0.6, lsd_n_bins=1024, bFLD=False, max_images=2, context=None, conv_seed_order=None):
        p = _lib.default_params()
        lp = p.line
        lp.lsd_nfeatures, lp.min_line_length, lp.lsd_refine = int(lsd_nfeatures), float(llength_th), int(lsd_refine)
        lp.lsd_scale, lp.lsd_sigma_scale, lp.lsd_quant, lp.lsd_ang_th = float(lsd_scale), float(lsd_sigma_scale), float(lsd_quant), float(lsd_ang_th)
        lp.lsd_log_eps, lp.lsd_density_th, lp.lsd_n_bins = float(lsd_log_eps), float(lsd_density_th), int(lsd_n_bins)
        if conv_seed_order is not None:      # convention C.9 (include/orbline_types.h); None = the library's default
            lp.conv_seed_order = int(conv_seed_order)
        self._params, self._max_images, self._ctx, self.bFLD = p, int(max_images), context, bool(bFLD)

    def _context(self, width, height, n_images):
        c = self._ctx
        if c is None or c.width != width or c.height != height or c.max_images < n_images:
            if c is not None:
                c.close()
            c = _lib.Context(self._params, width, height, max(n_images, self._max_images))
            self._ctx = c
        return c

    def __call__(self, image, mask=None):
        """image (H, W) uint8 -> (keylines[KEYLINE_DTYPE], descriptors (N, 32) uint8); mask ignored."""
        image = np.asarray(image)
        if self.bFLD or image.size == 0:
            return np.zeros(0, KEYLINE_DTYPE), np.zeros((0, DESC_BYTES), np.uint8)
        kls, desc, counts = self.extract_batch(image[None])
        n = int(counts[0])
        return kls[0, :n].copy(), desc[0, :n].copy()

    def extract_batch(self, images):
        images = np.asarray(images)
        if images.dtype != np.uint8 or images.ndim != 3:
            raise RuntimeError("Error, depth image!= 0")   # LSDDetector_custom.cpp:236-237
        images = np.ascontiguousarray(images)
        n, h, w = images.shape
        ctx = self._context(w, h, n)
        cap = ctx.line_capacity
        kls = np.zeros((n, cap), KEYLINE_DTYPE)
        desc = np.zeros((n, cap, DESC_BYTES), np.uint8)
        counts = np.zeros(n, np.int32)
        check(lib().olf_line_extract(ctx.handle, ptr(images), n, ptr(kls), ptr(desc), ptr(counts)), "olf_line_extract")
        return kls, desc, counts

    def compute(self, image, keylines):
        """BinaryDescriptor::compute(image, keylines, descriptors) on caller-supplied key lines."""
        image = np.ascontiguousarray(image)
        h, w = image.shape
        ctx = self._context(w, h, 1)
        cap = ctx.line_capacity
        if len(keylines) > cap:
            raise ValueError("more key lines than the extractor capacity")
        if len(keylines) == 0:
            return np.zeros((0, DESC_BYTES), np.uint8)   # "Error: keypoint list is empty" (binary_descriptor_custom.cpp:556-560)
        k = np.zeros((1, cap), KEYLINE_DTYPE)
        k[0, :len(keylines)] = keylines
        desc = np.zeros((1, cap, DESC_BYTES), np.uint8)
        counts = np.array([len(keylines)], np.int32)
        check(lib().olf_lbd_compute(ctx.handle, ptr(image[None].copy()), 1, ptr(k), ptr(counts), ptr(desc)), "olf_lbd_compute")
        return desc[0, :len(keylines)].copy()

    def debug_scaled(self, image=0):
        ctx = self._ctx
        buf = np.zeros(int(ctx.width * 1.5) * int(ctx.height * 1.5), np.uint8)
        ws, hs = C.c_int32(), C.c_int32()
        check(lib().olf_lsd_debug_scaled(ctx.handle, image, ptr(buf), C.byref(ws), C.byref(hs)), "olf_lsd_debug_scaled")
        return buf[:ws.value * hs.value].reshape(hs.value, ws.value).copy()
