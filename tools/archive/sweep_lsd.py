"""k_lsd_grow time vs images in flight (one wave per image): python tools/sweep_lsd.py 256 1024 4096 ..."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth
base = synth.stereo_batch(7000, 16, 1242, 375)
for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096]:
    imgs = np.tile(base, (n // 32 + 1, 1, 1))[:n].copy()
    ex = ola.Lineextractor(500, 0.025, max_images=n)
    ex.extract_batch(imgs)
    ex._ctx.profile(True)
    ex.extract_batch(imgs)
    ex._ctx.synchronize()
    p = ex._ctx.profile_read()
    print(n, {k: round(v[0], 2) for k, v in p.items() if v[1]}, "lsd_grow us/img %.1f" % (1e3 * p["lsd_grow"][0] / n), flush=True)
    del ex
