import sys, ctypes as C, numpy as np, os
sys.path.insert(0, '/root/repo')
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, h, seed = 640, 480, 71
Ws, Hs = 768, 576
imgs = synth.stereo_batch(seed, 2, w, h)
ex = ola.Lineextractor(0, 0.025, max_images=4)
ctx = ex._context(w, h, 4)
own = {}
for nw in (16, -2):
    if nw == -2:
        try: _lib.lib().olf_debug_ln_log(np.zeros(4, np.int32).ctypes.data_as(C.c_void_p), 0)
        except Exception: pass
    _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, nw, L if nw == -2 else 0), "w")
    ex.extract_batch(imgs)
    o = np.zeros(Ws * Hs, np.uint32)
    _lib.check(_lib.lib().olf_debug_lsd_owner(ctx.handle, 0, o.ctypes.data_as(C.c_void_p)), "o")
    own[nw] = o
a, b = own[16] >> 10, own[-2] >> 10
# only pixels claimed in the reference run are meaningful (undefined pixels hold garbage)
claimed = own[16] != 0xffffffff
diff = np.nonzero((a != b) & claimed & (a < 300000))[0]
print("pixels whose owner rank differs:", len(diff))
for p in diff[:12]:
    print("  pixel", p % Ws, p // Ws, "ref rank", a[p], "lane rank", b[p], "lane tag free" if own[-2][p] == 0xffffffff else "")

try:
    lg = np.zeros((65536, 4), np.int32)
    nl = _lib.lib().olf_debug_ln_log(lg.ctypes.data_as(C.c_void_p), 65536)
    print("log events", nl)
    lg = lg[:min(nl, 65536)]
    names = {1: "start", 2: "finish", 3: "fail"}
    for r in sorted(set(int(b[p]) for p in diff[:60])):
        print("rank", r, [(names[int(e[0])], int(e[2]), int(e[3])) for e in lg if e[1] == r])
except AttributeError:
    pass
for r in (1538,):
    pl = np.nonzero(b == r)[0]
    print("lane region", r, [(int(p % Ws), int(p // Ws), int(a[p])) for p in pl])
    pr = np.nonzero(a == r)[0]
    print("ref region", r, [(int(p % Ws), int(p // Ws)) for p in pr])
