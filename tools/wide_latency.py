"""Latency of the capacity path (csrc/lsd_wide.hip, 64-bit sort keys) beside the fast path: python tools/wide_latency.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth
def run(tag, w, h, n, **kw):
    ex = ola.Lineextractor(500, 0.025, max_images=n, **kw)
    imgs = np.stack([synth.stereo_pair(3 + i, w, h)[0] for i in range(min(n, 4))])
    imgs = np.tile(imgs, ((n + 3) // 4, 1, 1))[:n].copy()
    ex.extract_batch(imgs)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); ex.extract_batch(imgs); t.append(time.perf_counter() - t0)
    print("%-44s %4d images %8.2f ms per call (median of 5, host to host)" % (tag, n, 1e3 * sorted(t)[2]), flush=True)
    ex._ctx.close()
for so in (1, 0):
    run("KITTI 1242x375, 1024 bins, seed order %d" % so, 1242, 375, 2, conv_seed_order=so)
    run("KITTI 1242x375, 2048 bins (wide), seed order %d" % so, 1242, 375, 2, lsd_n_bins=2048, conv_seed_order=so)
    run("KITTI 1242x375, 2048 bins (wide), seed order %d" % so, 1242, 375, 64, lsd_n_bins=2048, conv_seed_order=so)
    run("KITTI 1242x375, 1024 bins, seed order %d" % so, 1242, 375, 64, conv_seed_order=so)
run("1920x1080, lsd_scale 1.2, 1024 bins", 1920, 1080, 2)
run("1920x1080, lsd_scale 2.0 (wide: 8.3 M pixels)", 1920, 1080, 2, lsd_scale=2.0)
