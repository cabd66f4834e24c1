"""One pair / few pairs through olf_stereo_frames (host buffers in and out) by workgroups per image of the growth kernel: python tools/ab_groups.py"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
p = _lib.default_params()
for n in (1, 2, 8, 32):
    fe = ola.StereoFrontEnd(p, 1242, 375, max_pairs=n)
    imgs = synth.stereo_batch(11, n, 1242, 375)
    ref = None
    for groups in (1, 2, 4, 1, 2, 4):
        _lib.check(_lib.lib().olf_debug_lsd_groups(fe.ctx.handle, groups), "groups")
        out = fe.frames(imgs)
        ts = []
        for _ in range(15):
            t = time.perf_counter(); out = fe.frames(imgs); ts.append(time.perf_counter() - t)
        kl = [out.pair(i)["mvKeys_Line"].tobytes() for i in range(n)]
        if ref is None: ref = kl
        print("%3d pairs, %d groups: median %.2f ms, min %.2f ms per call; lines identical to groups=1: %s" % (n, groups, 1e3 * np.median(ts), 1e3 * min(ts), kl == ref), flush=True)
