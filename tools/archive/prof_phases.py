"""Cycles per phase of a k_lsd_grow iteration (library built with -DOLF_TIMING2): python tools/prof_phases.py [images]"""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
imgs = synth.stereo_batch(7000, 16, 1242, 375)
imgs = np.tile(imgs, (n // 32 + 1, 1, 1))[:n].copy()
ex = ola.Lineextractor(500, 0.025, max_images=n)
for it in range(2):
    k, d, c = ex.extract_batch(imgs)
out = np.zeros(64, np.int32)
_lib.lib().olf_debug_status(ex._ctx.handle, out.ctypes.data_as(C.c_void_p))
t = out[16:32].view(np.int64)
names = ["ring+address", "gradient gather", "table lookups", "accept chain", "commit"]
print("images", n, "iterations", t[5], " ".join("%s %.0f" % (nm, t[i] / max(t[5], 1)) for i, nm in enumerate(names)), "cycles per iteration")
