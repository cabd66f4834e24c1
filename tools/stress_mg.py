"""The several-workgroups-per-image growth under UNEVEN load: one thread runs single-pair calls (2 growth groups + 8 sort groups per image, cross-CU agent-scope traffic)
and compares every result with the first one, while another thread keeps the chip busy with 256-pair batches on a context of its own.  python tools/stress_mg.py [seconds]"""
import sys, os, time, threading, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
p = _lib.default_params()
W, H = 1242, 375
stop = threading.Event()
stats = {"pair_calls": 0, "mismatch": 0, "batches": 0}


def load():
    fe = ola.StereoFrontEnd(p, W, H, max_pairs=256)
    imgs = np.tile(synth.stereo_batch(500, 16, W, H), (16, 1, 1))
    while not stop.is_set():
        fe.frames(imgs); stats["batches"] += 1


def probe(seed):
    fe = ola.StereoFrontEnd(p, W, H, max_pairs=1)
    imgs = synth.stereo_batch(seed, 1, W, H)
    ref = None
    while not stop.is_set():
        f = fe.frames(imgs)
        sig = (f.mvKeys_Line.tobytes(), f.mDescriptors_Line.tobytes(), f.mvKeysRight_Line.tobytes(), f.line_matches_12.tobytes(), f.mvKeys.tobytes(), f.mvuRight.tobytes())
        if ref is None: ref = sig
        elif sig != ref: stats["mismatch"] += 1
        stats["pair_calls"] += 1


# the reference of each probe is its first result, taken before the load starts
ths = [threading.Thread(target=probe, args=(900 + k,)) for k in range(2)]
for t in ths: t.start()
time.sleep(2.0)
tl = threading.Thread(target=load); tl.start()
time.sleep(secs)
stop.set()
for t in ths + [tl]: t.join()
print("STRESS %s: %d single-pair calls beside %d 256-pair batches, %d results differ from the unloaded first call" % ("OK" if stats["mismatch"] == 0 and stats["pair_calls"] > 20 else "FAILED", stats["pair_calls"], stats["batches"], stats["mismatch"]))
