"""Commit-wave trace of both groups of image 0 (library built with -DOLF_MW_PROF -DOLF_MW_TRACE): python tools/prof_mw_trace.py
What each group's head is waiting for, over time."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
imgs = synth.stereo_batch(7000, 1, 1242, 375)
ex = ola.Lineextractor(500, 0.025, max_images=4)
ctx = ex._context(1242, 375, 4)
_lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, 16, 0), "waves")
_lib.check(_lib.lib().olf_debug_lsd_groups(ctx.handle, 2), "groups")
Ps = 1490 * 450
# zero the trace area (owner words of image 3) by running four images once is not possible with 2: the area is hipMalloc'd garbage -> read, and only trust record counts that are sane
ex.extract_batch(imgs)
z = np.zeros(Ps, np.uint32)
_lib.check(_lib.lib().olf_debug_lsd_owner(ctx.handle, 3, z.ctypes.data_as(C.c_void_p)), "owner")
tr = z.view(np.int32)
names = {0: "EMPTY", 1: "READY", 2: "PARKED", 3: "GROWING", 4: "DONE", 5: "DEAD", -1: "-"}
for g in (0, 1):
    t = tr[g * 300000:]
    k = int(t[0])
    if not (0 < k <= 29000):
        print("group", g, "record count", k, "(garbage: the area was not zero before the first launch -- run again)"); continue
    r = t[10:10 + 10 * k].reshape(k, 10)
    # time by what the head was doing: consecutive polls with the same head index
    acc = {}
    for i in range(1, k):
        dt = r[i][0] - r[i - 1][0]
        key = names.get(int(r[i - 1][4]), "?") + (" (behind the other group)" if r[i - 1][4] == 4 and r[i - 1][3] >= r[i - 1][8] else "")
        acc[key] = acc.get(key, 0) + dt
    tot = sum(acc.values())
    print("group %d: %d polls, %.0f us; time by the state of the head entry: %s" % (g, k, tot / 100.0, ", ".join("%s %.0f us (%.0f%%)" % (kk, v / 100.0, 100.0 * v / tot) for kk, v in sorted(acc.items(), key=lambda kv: -kv[1]))))
    for i in range(0, k, max(1, k // 40)):
        tt, h, tl, hr, hs, moved, cnt, dn, omin, blk = r[i]
        print("  t %7.1f us head %6d tail %6d (%4d) head rank %6d %-7s blocker %6d | first 64: grow %2d ready %2d parked %2d done %2d | dispNext %6d others' wm %d" % (
            tt / 100.0, h, tl, tl - h, hr, names.get(int(hs), hs), blk, cnt & 255, (cnt >> 8) & 255, (cnt >> 16) & 255, (cnt >> 24) & 255, dn, omin))
