"""Commit trace of group 0 of image 0 (library built with -DOLF_MW_PROF -DOLF_MW_TRACE): python tools/prof_mw_trace.py [groups]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
imgs = synth.stereo_batch(7000, 1, 1242, 375)
grp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ex = ola.Lineextractor(500, 0.025, max_images=4)
ctx = ex._context(1242, 375, 4)
_lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, 16, 0), "waves")
_lib.check(_lib.lib().olf_debug_lsd_groups(ctx.handle, grp), "groups")
Ps = 1490 * 450
z = np.zeros(Ps, np.uint32)
# the trace area starts as whatever the owner words were: clear the record count by running once and reading (first launch: hipMalloc'd memory is not zero)
ex.extract_batch(imgs)
_lib.check(_lib.lib().olf_debug_lsd_owner(ctx.handle, 3, z.ctypes.data_as(C.c_void_p)), "owner")
tr = z.view(np.int32)
k = int(tr[0]); print("records", k)
r = tr[10:10 + 10 * k].reshape(k, 10)
names = {0: "EMPTY", 1: "READY", 2: "PARKED", 3: "GROWING", 4: "DONE", 5: "DEAD", -1: "-"}
# time the head spends per (rank): consecutive records with the same head rank
last = None; t_in = 0
stalls = []
for i in range(k):
    t, h, tl, hr, hs, hn, cnt, run, omin, blk = r[i]
    if last is None or hr != last[0]:
        if last is not None: stalls.append((t - last[1], last[0], last[2], last[3], last[4]))
        last = (hr, t, hs, hn, blk)
stalls.sort(reverse=True)
print("total time (us): %.0f; longest stays of one seed at the head (us, rank, state when it arrived, pixels then, blocker):" % (r[-1][0] / 100.0))
for d, hr, hs, hn, blk in stalls[:40]: print("  %7.1f us  rank %6d  %s  n %d  blocker %d" % (d / 100.0, hr, names.get(hs, hs), hn, blk))
tot = sum(s[0] for s in stalls)
print("sum of the 40 longest: %.0f us of %.0f" % (sum(s[0] for s in stalls[:40]) / 100.0, tot / 100.0))
# occupancy samples
for i in range(0, k, max(1, k // 60)):
    t, h, tl, hr, hs, hn, cnt, run, omin, blk = r[i]
    print("t %7.1f us head %6d tail %6d (%4d in buffer) head rank %6d %-7s n %4d | of the first 64: growing %2d ready %2d parked %2d done %2d | committed %2d omin %d" % (
        t / 100.0, h, tl, tl - h, hr, names.get(hs, hs), hn, cnt & 255, (cnt >> 8) & 255, (cnt >> 16) & 255, (cnt >> 24) & 255, run, omin))
