"""Soak: many synthetic stereo pairs (several seeds / sizes / image statistics) through the fused GPU path against the CPU oracle.
python tools/soak.py [pairs_per_config]   -- prints one line per mismatching pair and a summary; exit code 1 on any mismatch."""
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth
import oracle_lib as oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0; total = 0; t0 = time.time()
for (w, h, nf, nl, fx, bf) in [(1242, 375, 2000, 500, 718.856, 386.1448), (640, 480, 1000, 200, 435.2047, 47.9064), (752, 480, 1200, 300, 435.2047, 47.9064)]:
    p = oracle.full_params(nf, nl, fx, bf)
    fe = ola.StereoFrontEnd(p, w, h, max_pairs=N)
    for variant in range(3):
        imgs = synth.stereo_batch(1000 * variant + w, N, w, h)
        rng = np.random.default_rng(variant)
        if variant == 1:      # add sensor noise
            imgs = np.clip(imgs.astype(np.int16) + rng.integers(-6, 7, imgs.shape), 0, 255).astype(np.uint8)
        if variant == 2:      # low contrast + brightness ramp
            ramp = np.linspace(0, 60, w).astype(np.int16)[None, None, :]
            imgs = np.clip(imgs.astype(np.int16) // 2 + 40 + ramp, 0, 255).astype(np.uint8)
        f = fe.frames(imgs)
        for i in range(N):
            g = f.pair(i)
            o = oracle.stereo_points(imgs[2 * i], imgs[2 * i + 1], p)
            ol, orr = oracle.line_extract(imgs[2 * i], p.line), oracle.line_extract(imgs[2 * i + 1], p.line)
            m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], w, h, p.stereo)
            checks = {
                "kpsL": np.array_equal(g["mvKeys"], o["kpsL"]), "descL": np.array_equal(g["mDescriptors"], o["descL"]),
                "kpsR": np.array_equal(g["mvKeysRight"], o["kpsR"]), "descR": np.array_equal(g["mDescriptorsRight"], o["descR"]),
                "uRight": np.array_equal(g["mvuRight"].view(np.uint32), o["uRight"].view(np.uint32)),
                "depth": np.array_equal(g["mvDepth"].view(np.uint32), o["depth"].view(np.uint32)),
                "klsL": g["mvKeys_Line"].tobytes() == np.ascontiguousarray(ol["kls"]).tobytes(),
                "klsR": g["mvKeysRight_Line"].tobytes() == np.ascontiguousarray(orr["kls"]).tobytes(),
                "ldescL": np.array_equal(g["mDescriptors_Line"], ol["desc"]), "ldescR": np.array_equal(g["mDescriptorsRight_Line"], orr["desc"]),
                "lmatch": np.array_equal(g["line_matches_12"], m), "ldisp": np.array_equal(g["mvDisparity_l"].view(np.uint32), disp.view(np.uint32)),
                "le": np.array_equal(g["mvle_l"].view(np.uint64), le.view(np.uint64)),
            }
            total += 1
            if not all(checks.values()):
                bad += 1
                print(f"MISMATCH {w}x{h} variant {variant} pair {i}: " + ",".join(k for k, v in checks.items() if not v), flush=True)
        print(f"{w}x{h} variant {variant}: {N} pairs, kps/img {np.mean(f.N):.0f}, lines/img {np.mean(f.N_l):.0f}, mismatching so far {bad}, {time.time() - t0:.0f} s", flush=True)
print(f"SOAK {'FAILED' if bad else 'OK'}: {total - bad}/{total} pairs bit-identical")
sys.exit(1 if bad else 0)
