"""Cycles per phase of the multi-wave growth kernel, summed over the waves of image 0 (library built with -DOLF_MW_PROF):
python tools/prof_mw.py [images] [waves ...]   (OLF_PROF_GROUPS=1,2,4: workgroups per image to run each setting with)"""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
base = synth.stereo_batch(7000, 16, 1242, 375)
imgs = np.tile(base, (n // 32 + 1, 1, 1))[:n].copy()
names = ["commit", "pick", "dispatch", "prologue", "gather", "chain", "claim", "finish", "idle", "runs", "iters", "fails", "f_inval", "f_contest", "f_older", "f_dup"]
groups = [int(a) for a in os.environ.get("OLF_PROF_GROUPS", "1").split(",")]
for nw, grp in [(int(a), g) for a in (sys.argv[2:] or [1, 4, 16]) for g in groups]:
    ex = ola.Lineextractor(500, 0.025, max_images=n)
    ctx = ex._context(1242, 375, n)
    _lib.check(_lib.lib().olf_debug_lsd_waves(ctx.handle, nw, 0), "waves")
    _lib.check(_lib.lib().olf_debug_lsd_groups(ctx.handle, grp), "groups")
    ex.extract_batch(imgs)
    z = np.zeros(256, np.int32)
    # reset the counters: the status block is device memory; read, then run once more and take the difference
    _lib.lib().olf_debug_status_n(ctx.handle, z.ctypes.data_as(C.c_void_p), 256)
    t0 = z[16:48].view(np.int64).copy(); l0 = z[100:132].view(np.int64).copy()
    ex.extract_batch(imgs)
    _lib.lib().olf_debug_status_n(ctx.handle, z.ctypes.data_as(C.c_void_p), 256)
    for label, t in (("whole launch", z[16:48].view(np.int64) - t0), ("behind seed 40960", z[100:132].view(np.int64) - l0)):
      tot = t[:9].sum()
      print(label + ": " +"images %d waves %d groups %d: total %.2f Mcycles (per wave %.2f) | " % (n, nw, grp, tot / 1e6, tot / 1e6 / nw / grp) +
            " ".join("%s %.1f%%" % (names[i], 100.0 * t[i] / tot) for i in range(9)) +
            " | runs %d iters %d fails %d (inval %d contest %d older %d dup %d); cycles/iter: gather %.0f chain %.0f claim %.0f; per run: prologue %.0f finish %.0f pick %.0f commit %.0f" %
            (t[9], t[10], t[11], t[12], t[13], t[14], t[15], t[4] / t[10], t[5] / t[10], t[6] / t[10], t[3] / t[9], t[7] / t[9], t[1] / t[9], t[0] / t[9]), flush=True)
