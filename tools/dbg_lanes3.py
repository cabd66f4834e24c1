import sys, ctypes as C, numpy as np, os
sys.path.insert(0, '/root/repo')
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, h, seed = 640, 480, 71
imgs = synth.stereo_batch(seed, 2, w, h)
ex = ola.Lineextractor(0, 0.025, max_images=4)
ctx = ex._context(w, h, 4)
res = {}
for nw in (16, -2):
    _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, nw, L if nw == -2 else 0), "w")
    ex.extract_batch(imgs)
    for im in range(2):
        sn = np.zeros((20000, 2), np.int32); ang = np.zeros(20000); cnt = C.c_int32()
        _lib.check(_lib.lib().olf_debug_lsd_regions(ctx.handle, im, sn.ctypes.data_as(C.c_void_p), ang.ctypes.data_as(C.c_void_p), 20000, C.byref(cnt)), "r")
        res[(nw, im)] = (sn[:cnt.value, 1].copy(), ang[:cnt.value].copy())
for im in range(2):
    a, b = res[(16, im)], res[(-2, im)]
    print("image", im, "regions", len(a[0]), len(b[0]))
    m = min(len(a[0]), len(b[0]))
    bad = [i for i in range(m) if a[0][i] != b[0][i] or a[1][i] != b[1][i]]
    print("  first mismatches", bad[:5])
    for i in bad[:3]:
        print("   region", i, "ref n,ang", a[0][i], a[1][i], " lane n,ang", b[0][i], b[1][i], " next ref n:", a[0][i + 1:i + 4], "next lane n:", b[0][i + 1:i + 4])
