"""seed-sort variant probe: python tools/ss_probe.py mode n_images [distinct]  -- time of the line path and the status word"""
import sys, time, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
mode, n = int(sys.argv[1]), int(sys.argv[2])
distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 32
base = synth.stereo_batch(7000, 16, 1242, 375)[:distinct]
imgs = np.tile(base, (n // distinct + 1, 1, 1))[:n].copy()
ex = ola.Lineextractor(500, 0.025, max_images=n, conv_seed_order=1)
ctx = ex._context(1242, 375, n)
_lib.check(_lib.lib().olf_debug_seed_sort_mode(ctx.handle, mode), "mode")
for it in range(2):
    t = time.time()
    try:
        k, d, c = ex.extract_batch(imgs); msg = "mean lines %.1f" % c.mean()
    except Exception as e:
        msg = "ERROR " + str(e)[:60]
    print("mode", mode, "images", n, "distinct", distinct, "iter", it, "%.3f s" % (time.time() - t), msg, flush=True)
