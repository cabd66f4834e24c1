"""A/B of library variants on the small-batch latency of the fused entry (host buffers in and out): python tools/ab_pair_latency.py base build/variants/x.so ...
Every library runs in its own process (OLF_LIB_PATH), twice in alternation; prints the median milliseconds per call for 1, 8 and 32 pairs."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
p = _lib.default_params()
out = []
for n in (1, 8, 32):
    fe = ola.StereoFrontEnd(p, 1242, 375, max_pairs=n)
    imgs = synth.stereo_batch(11, n, 1242, 375)
    fe.frames(imgs); fe.frames(imgs)
    ts = []
    for _ in range(15):
        t = time.perf_counter(); fe.frames(imgs); ts.append(time.perf_counter() - t)
    out.append("%%d pairs %%.2f ms" %% (n, 1e3 * float(np.median(ts))))
    fe.ctx.close()
print(" | ".join(out), flush=True)
''' % ROOT
libs = sys.argv[1:] or ["base"]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "base":
            env["OLF_LIB_PATH"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        else:
            env.pop("OLF_LIB_PATH", None)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        print("%-28s %s" % (lib, r.stdout.decode().strip() or r.stderr.decode()[-300:]), flush=True)
