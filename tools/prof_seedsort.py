"""Cycles per phase of k_lsd_seedsort for image 0 (library built with -DOLF_SS_PROF): python tools/prof_seedsort.py [images]"""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
base = synth.stereo_batch(7000, 16, 1242, 375)
imgs = np.tile(base, (n // 32 + 1, 1, 1))[:n].copy()
ex = ola.Lineextractor(500, 0.025, max_images=n, conv_seed_order=1)
ex.extract_batch(imgs)
ex.extract_batch(imgs)
z = np.zeros(64, np.int32)
_lib.lib().olf_debug_status(ex._ctx.handle, z.ctypes.data_as(C.c_void_p))
t = z[16:16 + 30].view(np.int64)
names = ["part_mem", "part_lds", "equal", "leaf", "load", "pivot", "other"]
tot = t[:7].sum()
print("images %d: image 0 total %.2f Mcycles | " % (n, tot / 1e6) + " ".join("%s %.1f%%" % (names[i], 100.0 * t[i] / tot) for i in range(7)))
print("  partitions mem %d (%d visits, %.0f cycles per 64) lds %d (%d visits, %.0f cycles per 64); equal ranges %d (%d elements, %.0f cycles per 64); leaves %d (%.0f cycles each); lds loads %d (%.0f cycles each); pivot %.0f cycles each" %
      (t[7], t[12], 64.0 * t[0] / max(t[12], 1), t[8], t[13], 64.0 * t[1] / max(t[13], 1), t[9], t[14], 64.0 * t[2] / max(t[14], 1), t[10], t[3] / max(t[10], 1), t[11], t[4] / max(t[11], 1), t[5] / max(t[7] + t[8], 1)))
