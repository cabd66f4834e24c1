import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
imgs = synth.stereo_batch(100, n // 2, 1242, 375)
ex = ola.ORBextractor(2000, 1.2, 8, 20, 7, max_images=n)
for it in range(3):
    t = time.time(); k, d, c = ex.extract_batch(imgs); dt = time.time() - t
    print("iter", it, "images", n, "sec", dt, "mean kps", c.mean())
