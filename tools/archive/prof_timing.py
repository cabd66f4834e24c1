import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
imgs = synth.stereo_batch(7000, 16, 1242, 375)
imgs = np.tile(imgs, (n // 32 + 1, 1, 1))[:n].copy()
ex = ola.Lineextractor(500, 0.025, max_images=n)
for it in range(2):
    k, d, c = ex.extract_batch(imgs)
out = np.zeros(64, np.int32)
_lib.lib().olf_debug_status(ex._ctx.handle, out.ctypes.data_as(C.c_void_p))
t = out[16:32].view(np.int64)
names = ["t_seed", "t_small", "t_big", "t_rect", "n_small", "n_big", "it_small", "it_big"]
print(dict(zip(names, t.tolist())))
tot = t[:4].sum()
print("cycles total %.1fM; seed %.1f%% small %.1f%% big %.1f%% rect %.1f%%" % (tot / 1e6, 100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot))
print("cycles/iter small %.0f big %.0f; per small region %.0f; seed-phase per region %.0f" % (t[1] / max(t[6], 1), t[2] / max(t[7], 1), t[1] / max(t[4], 1), t[0] / max(t[4] + t[5], 1)))
