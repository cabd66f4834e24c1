"""Latency of the fused stereo-frame entry for small batches (host buffers in, host buffers out): python tools/pair_latency.py"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
p = _lib.default_params()
for n in (1, 2, 4, 8, 32, 128):
    fe = ola.StereoFrontEnd(p, 1242, 375, max_pairs=n)
    imgs = synth.stereo_batch(11, n, 1242, 375)
    fe.frames(imgs)
    t = time.perf_counter()
    reps = 5
    for _ in range(reps): fe.frames(imgs)
    dt = (time.perf_counter() - t) / reps
    print("%4d pairs per call: %.1f ms per call, %.2f ms per pair" % (n, 1e3 * dt, 1e3 * dt / n), flush=True)
