"""Region-growing stage time (HIP events around k_lsd_grow / k_lsd_grow_mw) over batch size x waves per image:
python tools/grow_sweep.py [pairs ...]   (waves 0 = the one-wave agent of round 1)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 128, 512, 2048]
W, H = 1242, 375
p = _lib.default_params()
base = synth.stereo_batch(11, 32, W, H)
for n in sizes:
    fe = ola.StereoFrontEnd(p, W, H, max_pairs=n)
    imgs = np.tile(base, ((n + 31) // 32, 1, 1))[:2 * n].copy()
    row = []
    for nw in [int(x) for x in os.environ.get('OLF_SWEEP_NW', '0,1,2,4,8,16').split(',')]:
        _lib.check(_lib.lib().olf_debug_lsd_waves(fe.ctx.handle, nw, 0), "olf_debug_lsd_waves")
        fe.frames(imgs)
        fe.ctx.profile(True)
        reps = 3
        for _ in range(reps): fe.frames(imgs)
        fe.ctx.synchronize()
        prof = fe.ctx.profile_read()
        fe.ctx.profile(False)
        row.append("nw%-2d %8.2f" % (nw, prof["lsd_grow"][0] / reps))
    print("%5d pairs: lsd_grow ms per call: " % n + " | ".join(row), flush=True)
    del fe
