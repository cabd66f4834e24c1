import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
w, h = 640, 480
imgs = synth.stereo_batch(71, 2, w, h)
ex = ola.Lineextractor(0, 0.025, max_images=4)
for groups, waves, rob in [(2, 16, 512), (2, 1, 128), (4, 16, 512)]:
    ctx = ex._context(w, h, 4)
    _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, waves, rob), "w")
    _lib.check(_lib.lib().olf_debug_lsd_groups(ctx.handle, groups), "g")
    try:
        kls, desc, counts = ex.extract_batch(imgs)
        print(groups, waves, rob, "ok", counts)
    except Exception as e:
        print(groups, waves, rob, "FAILED", str(e)[:80])
    out = np.zeros(256, np.int32)
    _lib.lib().olf_debug_status_n(ctx.handle, _lib.ptr(out), 256)
    if out[0] & 16:
      for g in range(4):
        d = out[64 + g * 24: 88 + g * 24]
        print("  grp %d: head %d tail %d dispNext %d wm %d omin %d wml %d | head: state %d rank %d inval %d block %d | nkeys %d n %d | gwm %s ginval %d abort %d | locks %d %d owner(seed) %x seed %x" % (
            d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11], d[12], d[13], d[14:18], d[18], d[19], d[20], d[21], d[22] & 0xffffffff, d[23] & 0xffffffff))
      break
