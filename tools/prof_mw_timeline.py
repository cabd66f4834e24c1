"""When the dispatcher of image 0 crosses every 4096th seed rank (library built with -DOLF_MW_PROF): python tools/prof_mw_timeline.py"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
imgs = synth.stereo_batch(7000, 1, 1242, 375)
for grp in [int(a) for a in os.environ.get("OLF_PROF_GROUPS", "1,2").split(",")]:
    ex = ola.Lineextractor(500, 0.025, max_images=2)
    ctx = ex._context(1242, 375, 2)
    _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, 16, 0), "waves")
    _lib.check(_lib.lib().olf_debug_lsd_groups(ctx.handle, grp), "groups")
    ex.extract_batch(imgs)
    z = np.zeros(256, np.int32)
    _lib.lib().olf_debug_status_n(ctx.handle, z.ctypes.data_as(C.c_void_p), 256)
    t = z[160:190] / 100.0      # us
    n = int(np.max(np.nonzero(t)[0])) + 1 if t.any() else 0
    print("groups %d: kernel end (us, per group) %s" % (grp, [round(v / 100.0) for v in z[250:250 + grp]]))
    print("  rank 4096*k reached at (us):   " + " ".join("%5d" % round(v) for v in t[:n]))
    print("  entries inserted by then:      " + " ".join("%5d" % v for v in z[190:190 + n]))
    print("  region runs started by then:   " + " ".join("%5d" % v for v in z[220:220 + n]), flush=True)
