"""lsd_refine = LSD_REFINE_ADV (k_lsd_grow<2>) timing over batch sizes: python tools/adv_timing.py [refine] [sizes ...]"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth
refine = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sizes = [int(a) for a in sys.argv[2:]] or [64, 1024, 4096]
base = synth.stereo_batch(7000, 16, 1242, 375)
for n in sizes:
    imgs = np.tile(base, (n // 32 + 1, 1, 1))[:n].copy()
    ex = ola.Lineextractor(500, 0.025, lsd_refine=refine, max_images=n)
    ex.extract_batch(imgs)
    ex._ctx.profile(True)
    ex.extract_batch(imgs); ex._ctx.synchronize()
    prof = ex._ctx.profile_read(); ex._ctx.profile(False)
    print("refine %d, %5d images: lsd_grow %.2f ms, lsd_front %.2f ms" % (refine, n, prof["lsd_grow"][0], prof["lsd_front"][0]), flush=True)
    ex._ctx.close()
