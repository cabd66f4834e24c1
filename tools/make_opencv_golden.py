#!/usr/bin/env python3
"""Pins the CPU oracle's restatement of the un-vendored OpenCV primitives against a real OpenCV.

Run this ONCE on any machine that has OpenCV 3.4.x with Python bindings (3.4.0-3.4.5 still ship cv::createLineSegmentDetector; the image
this repository is built in has no OpenCV at all, so the fixtures cannot be made there):

    python tools/make_opencv_golden.py            # writes tests/golden/opencv34_*.npz

and commit the files.  tests/test_oracle_cpu.py::test_oracle_against_opencv_golden then compares the oracle with them (it is skipped while
they are absent).  The inputs are this repository's seeded synthetic images, so nothing but the expected outputs is stored.  Each file records
cv2.__version__; a mismatch in one primitive tells which convention of DESIGN.md section 2 (C.9 seed order, C.10 INTER_LINEAR vs _EXACT, C.11
Gaussian taps, C.12 small-matrix gemm, C.13 initUndistortRectifyMap, C.14 LSD refinement) the linked OpenCV follows.  Round 4 added: LSD with
refine 1 / 2, the 8-bit Gaussian taps (impulse responses), BFMatcher.knnMatch with ties, CV_32F 3x3 gemm in both forms the matcher uses,
LineIterator counts (through cv2.line), the EuRoC rectification maps and -- with opencv_contrib -- line_descriptor's KeyLines + LBD descriptors.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import cv2
    from orb_line_slam_amd import synth
    out = os.path.join(ROOT, "tests", "golden")
    ver = cv2.__version__
    left, right = synth.stereo_pair(11, 320, 240)
    big, _ = synth.stereo_pair(12, 640, 480)
    meta = dict(cv_version=ver)
    # A.2 resize INTER_LINEAR (pyramid step 1/1.2) and LSD's x1.2 upsampling; A.10 variant INTER_LINEAR_EXACT when the build has it
    w1, h1 = int(round(320 / 1.2)), int(round(240 / 1.2))
    r = dict(meta, down=cv2.resize(left, (w1, h1), interpolation=cv2.INTER_LINEAR), up=cv2.resize(left, None, fx=1.2, fy=1.2, interpolation=cv2.INTER_LINEAR))
    if hasattr(cv2, "INTER_LINEAR_EXACT"):
        r["up_exact"] = cv2.resize(left, None, fx=1.2, fy=1.2, interpolation=cv2.INTER_LINEAR_EXACT)
    np.savez_compressed(os.path.join(out, "opencv34_resize.npz"), **r)
    # A.3 GaussianBlur 7x7 sigma 2 (ORB), 5x5 sigma 1 (LBD), 7x7 sigma 0.6 (LSD)
    np.savez_compressed(os.path.join(out, "opencv34_blur.npz"), **dict(meta, s2=cv2.GaussianBlur(left, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101),
                        s1=cv2.GaussianBlur(left, (5, 5), 1, 1), s06=cv2.GaussianBlur(left, (7, 7), 0.6, 0.6)))
    # A.4 FAST-9/16 with non-maximum suppression at the two thresholds of the path
    f = {}
    for th in (20, 7):
        det = cv2.FastFeatureDetector_create(threshold=th, nonmaxSuppression=True, type=cv2.FAST_FEATURE_DETECTOR_TYPE_9_16)
        k = det.detect(left, None)
        f["th%d" % th] = np.array([(p.pt[0], p.pt[1], p.response) for p in k], np.float32).reshape(-1, 3)
    np.savez_compressed(os.path.join(out, "opencv34_fast.npz"), **dict(meta, **f))
    # A.5 fastAtan2 on a grid of integer arguments (the path's arguments are integer moments / gradients)
    ys, xs = np.meshgrid(np.arange(-64, 65, dtype=np.float32), np.arange(-64, 65, dtype=np.float32), indexing="ij")
    np.savez_compressed(os.path.join(out, "opencv34_atan2.npz"), **dict(meta, y=ys, x=xs, deg=np.array([[cv2.fastAtan2(float(a), float(b)) for a, b in zip(ry, rx)]
                                                                                                       for ry, rx in zip(ys, xs)], np.float32)))
    # A.9 Sobel 3x3 to int16
    np.savez_compressed(os.path.join(out, "opencv34_sobel.npz"), **dict(meta, dx=cv2.Sobel(left, cv2.CV_16S, 1, 0, ksize=3), dy=cv2.Sobel(left, cv2.CV_16S, 0, 1, ksize=3)))
    # A.7 LSD exactly as LSDDetectorC::detectImpl sets it up (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:246-253)
    try:
        lsd = cv2.createLineSegmentDetector(0, 1.2, 0.6, 2.0, 22.5, 1.0, 0.6, 1024)
        d = dict(meta)
        for name, img in (("small", left), ("big", big)):
            lines = lsd.detect(img)[0]
            d[name] = np.zeros((0, 4), np.float32) if lines is None else lines.reshape(-1, 4).astype(np.float32)
        np.savez_compressed(os.path.join(out, "opencv34_lsd.npz"), **d)
    except cv2.error as e:
        print("this OpenCV build has no LineSegmentDetector (removed in 3.4.6-3.4.15 / 4.1.0-4.5.3):", e)
    # C.14: LSD with LSD_REFINE_STD (1) and LSD_REFINE_ADV (2) -- refine / reduce_region_radius / rect_improve / rect_nfa are restated from memory in the oracle
    try:
        d = dict(meta)
        for refine in (1, 2):
            lsd_r = cv2.createLineSegmentDetector(refine, 1.2, 0.6, 2.0, 22.5, 1.0, 0.6, 1024)
            for name, img in (("small", left), ("big", big)):
                lines = lsd_r.detect(img)[0]
                d["%s_refine%d" % (name, refine)] = np.zeros((0, 4), np.float32) if lines is None else lines.reshape(-1, 4).astype(np.float32)
        np.savez_compressed(os.path.join(out, "opencv34_lsd_refine.npz"), **d)
    except cv2.error as e:
        print("no LineSegmentDetector in this build:", e)
    # C.11: the 8-bit Gaussian taps themselves (getGaussianKernel is double; the fixed-point taps are what GaussianBlur on CV_8U uses: their image is the
    # blur of a unit impulse x 256 -- a 15 x 15 image with one pixel of 255 does not saturate any tap)
    imp = np.zeros((15, 15), np.uint8); imp[7, 7] = 255
    np.savez_compressed(os.path.join(out, "opencv34_gauss_taps.npz"), **dict(meta, k7s2=cv2.getGaussianKernel(7, 2.0), k5s1=cv2.getGaussianKernel(5, 1.0), k7s06=cv2.getGaussianKernel(7, 0.6),
                        imp_s2=cv2.GaussianBlur(imp, (7, 7), 2, 2), imp_s1=cv2.GaussianBlur(imp, (5, 5), 1, 1), imp_s06=cv2.GaussianBlur(imp, (7, 7), 0.6, 0.6)))
    # A.10 BFMatcher(NORM_HAMMING).knnMatch(k = 2): what matchNNR / match of src/LineMatcher.cpp:42-132 are built on, incl. exact ties (duplicated train rows)
    rng = np.random.default_rng(77)
    q = rng.integers(0, 256, (300, 32), dtype=np.uint8); t = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    t[40] = t[3]; t[200] = t[3]; q[7] = t[3]; q[8] = t[3] ^ np.uint8(1)        # ties for the best and for the second best
    knn = cv2.BFMatcher(cv2.NORM_HAMMING).knnMatch(q, t, k=2)
    np.savez_compressed(os.path.join(out, "opencv34_knn.npz"), **dict(meta, q=q, t=t, idx=np.array([[m.trainIdx for m in r] for r in knn], np.int32),
                        dist=np.array([[m.distance for m in r] for r in knn], np.float32)))
    # C.12 cv::Mat products of CV_32F operands as the ORBmatcher searches write them: R * x + t (gemm, inner length 3) and -R.t() * t (transposed operand)
    R = rng.standard_normal((64, 3, 3)).astype(np.float32); x = (rng.standard_normal((64, 3, 1)) * 30).astype(np.float32); tt = rng.standard_normal((64, 3, 1)).astype(np.float32)
    rx_t = np.stack([cv2.gemm(R[i], x[i], 1.0, tt[i], 1.0) for i in range(64)])
    mrt_t = np.stack([cv2.gemm(R[i], tt[i], -1.0, None, 0.0, flags=cv2.GEMM_1_T) for i in range(64)])
    np.savez_compressed(os.path.join(out, "opencv34_gemm.npz"), **dict(meta, R=R, x=x, t=tt, Rx_plus_t=rx_t, minus_Rt_t=mrt_t))
    # cv::LineIterator(img, Point2f -> Point, Point2f -> Point).count (KeyLine.numOfPixels, LSDDetector_custom.cpp:294-295): the Python bindings of 3.4 do not
    # expose the class; an 8-connected cv2.line of thickness 1 draws exactly the iterator's pixels, so its pixel count is li.count
    segs = (rng.random((200, 4)) * np.array([319, 239, 319, 239])).astype(np.float32)
    cnt = []
    for s in segs:
        cv = np.zeros((240, 320), np.uint8)
        cv2.line(cv, (int(round(float(s[0]))), int(round(float(s[1])))), (int(round(float(s[2]))), int(round(float(s[3])))), 255, 1, 8)
        cnt.append(int(np.count_nonzero(cv)))
    np.savez_compressed(os.path.join(out, "opencv34_lineiterator.npz"), **dict(meta, segs=segs, count=np.array(cnt, np.int32)))
    # C.13 initUndistortRectifyMap with the EuRoC calibration of Examples/PL/PL_EuRoC.yaml (left camera)
    K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]]); D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0])
    Rr = np.array([[0.999966347530033, -0.001422739138722922, 0.008079580483432283], [0.001365741834644127, 0.9999741760894847, 0.007055629199258132],
                   [-0.008089410156878961, -0.007044357138835809, 0.9999424675829176]])
    Pp = np.array([[435.2046959714599, 0, 367.4517211914062, 0], [0, 435.2046959714599, 252.2008514404297, 0], [0, 0, 1, 0]])
    m1, m2 = cv2.initUndistortRectifyMap(K, D, Rr, Pp[:3, :3], (752, 480), cv2.CV_32F)
    np.savez_compressed(os.path.join(out, "opencv34_rectify.npz"), **dict(meta, K=K, D=D, R=Rr, P=Pp, m1=m1, m2=m2))
    # opencv_contrib's line_descriptor (the code the reference's Thirdparty/line_descriptor was forked from): KeyLines and LBD descriptors of its own LSD
    # wrapper, when the bindings exist -- an end-to-end cross-check of make_keylines / lbd_compute (the fork changes options, not the arithmetic)
    if hasattr(cv2, "line_descriptor"):
        try:
            det = cv2.line_descriptor.LSDDetector_createLSDDetector()
            kl = det.detect(left, 2, 1)
            kl = [k for k in kl if k.octave == 0]
            bd = cv2.line_descriptor.BinaryDescriptor_createBinaryDescriptor()
            kl2, desc = bd.compute(left, kl)
            arr = np.array([(k.angle, k.class_id, k.octave, k.pt[0], k.pt[1], k.response, k.size, k.startPointX, k.startPointY, k.endPointX, k.endPointY,
                             k.sPointInOctaveX, k.sPointInOctaveY, k.ePointInOctaveX, k.ePointInOctaveY, k.lineLength, k.numOfPixels) for k in kl2], np.float64)
            np.savez_compressed(os.path.join(out, "opencv34_contrib_lbd.npz"), **dict(meta, keylines=arr, desc=np.asarray(desc, np.uint8)))
        except Exception as e:
            print("line_descriptor bindings present but unusable:", e)
    print("wrote tests/golden/opencv34_*.npz with OpenCV", ver)


if __name__ == "__main__":
    main()
