#!/usr/bin/env python3
"""Pins the CPU oracle's restatement of the un-vendored OpenCV primitives against a real OpenCV.

Run this ONCE on any machine that has OpenCV 3.4.x with Python bindings (3.4.0-3.4.5 still ship cv::createLineSegmentDetector; the image
this repository is built in has no OpenCV at all, so the fixtures cannot be made there):

    python tools/make_opencv_golden.py            # writes tests/golden/opencv34_*.npz

and commit the files.  tests/test_oracle_cpu.py::test_oracle_against_opencv_golden then compares the oracle with them (it is skipped while
they are absent).  The inputs are this repository's seeded synthetic images, so nothing but the expected outputs is stored.  Each file records
cv2.__version__; a mismatch in one primitive tells which convention of DESIGN.md section 2 (C.9 seed order, C.10 INTER_LINEAR vs _EXACT, C.11
Gaussian taps) the linked OpenCV follows.
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
    print("wrote tests/golden/opencv34_*.npz with OpenCV", ver)


if __name__ == "__main__":
    main()
