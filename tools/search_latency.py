"""Per-call latency of the native per-frame searches (one frame, 2000 ORB features): python tools/search_latency.py"""
import sys, time, copy, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import orb_line_slam_amd as ola
import oracle_lib as oracle
from test_search_gpu import _frames
last, cur = _frames(oracle)
m = ola.ORBmatcher(0.9, True)
rng = np.random.default_rng(0)
sel = np.flatnonzero(last.mp_valid)
mp = ola.MapPointView(last.mDescriptors[sel], last.mvKeysUn["x"][sel] + 3, last.mvKeysUn["y"][sel], last.mvKeysUn["x"][sel] - 20,
                      last.mvKeysUn["octave"][sel], rng.uniform(0.99, 1.0, len(sel)))
v0, o0 = cur.mp_valid.copy(), cur.mp_obs.copy()
def reset():
    np.copyto(cur.mp_valid, v0); np.copyto(cur.mp_obs, o0)
def timed(fn, reps=100):
    reset(); fn()
    tot = 0.0
    for _ in range(reps):
        reset()
        t = time.perf_counter(); fn(); tot += time.perf_counter() - t
    return 1e3 * tot / reps
proj = lambda: m.SearchByProjection(cur, last, 7, False)
local = lambda: m.SearchByProjection(cur, mp, 1.0)
tp, tl = timed(proj), timed(local)
reset(); np_ = proj()[0]; reset(); nl = local()[0]
print("map points projected: %d; SearchByProjection(Frame, Frame): %.2f ms per call, %d matches; SearchByProjection(Frame, local map of %d points): "
      "%.2f ms per call, %d matches (ctypes view packing included)" % (int(last.mp_valid.sum()), tp, np_, mp.n, tl, nl))
