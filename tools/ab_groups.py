"""One pair / few pairs through olf_stereo_frames (host buffers in and out) by workgroups per image of the growth kernel:
python tools/ab_groups.py [pairs ...]    (OLF_AB_SETTINGS="groups:rob,..." e.g. "1:512,2:512,2:1024,4:1024"; OLF_LSD_WS picks the window)"""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
p = _lib.default_params()
settings = [tuple(int(x) for x in s.split(":")) for s in os.environ.get("OLF_AB_SETTINGS", "1:512,2:512,4:512").split(",")]
for n in [int(a) for a in sys.argv[1:]] or [1, 8]:
    fe = ola.StereoFrontEnd(p, 1242, 375, max_pairs=n)
    imgs = synth.stereo_batch(11, n, 1242, 375)
    ref = None
    for groups, rob in settings * 2:
        _lib.check(_lib.lib().olf_debug_lsd_groups(fe.ctx.handle, groups), "groups")
        _lib.check(_lib.lib().olf_debug_lsd_waves(fe.ctx.handle, 16, rob), "waves")
        out = fe.frames(imgs)
        ts = []
        for _ in range(15):
            t = time.perf_counter(); out = fe.frames(imgs); ts.append(time.perf_counter() - t)
        kl = [out.pair(i)["mvKeys_Line"].tobytes() for i in range(n)]
        if ref is None: ref = kl
        print("ws %s %3d pairs, %d groups, rob %4d: median %.2f ms, min %.2f ms per call; lines identical: %s" % (os.environ.get("OLF_LSD_WS", "10"), n, groups, rob, 1e3 * np.median(ts), 1e3 * min(ts), kl == ref), flush=True)
