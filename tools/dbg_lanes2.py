import sys, ctypes as C, numpy as np, os
sys.path.insert(0, '/root/repo')
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
for (w, h, seed) in ((640, 480, 71), (1242, 375, 7000), (320, 240, 5)):
    imgs = synth.stereo_batch(seed, 2, w, h)
    ex = ola.Lineextractor(0, 0.025, max_images=4)
    ctx = ex._context(w, h, 4)
    out = {}
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    for nw in (16, -2):
        _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, nw, L if nw == -2 else 0), "w")
        k, d, c = ex.extract_batch(imgs)
        out[nw] = (k.copy(), c.copy())
    for i in range(4):
        a, b = out[16][0][i, :out[16][1][i]], out[-2][0][i, :out[-2][1][i]]
        if len(a) != len(b):
            print(w, h, "image", i, "counts differ", len(a), len(b)); continue
        bad = [j for j in range(len(a)) if a[j].tobytes() != b[j].tobytes()]
        print(w, h, "image", i, "lines", len(a), "mismatching", len(bad), bad[:8])
        for j in bad[:2]:
            print("   ref ", a[j][["startPointX", "startPointY", "endPointX", "endPointY", "numOfPixels"]])
            print("   lane", b[j][["startPointX", "startPointY", "endPointX", "endPointY", "numOfPixels"]])
