import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
so = int(sys.argv[2]) if len(sys.argv) > 2 else 1
imgs = synth.stereo_batch(7000, 16, 1242, 375)
imgs = np.tile(imgs, (n // 32 + 1, 1, 1))[:n].copy()
ex = ola.Lineextractor(500, 0.025, max_images=n, conv_seed_order=so)
for it in range(2):
    t = time.time(); k, d, c = ex.extract_batch(imgs); dt = time.time() - t
    print("iter", it, "images", n, "sec", dt, "mean lines", c.mean())
