"""Randomised parity fuzz: random image sizes, extractor / detector / stereo parameters and image statistics through the fused GPU
path against the CPU oracle.  python tools/fuzz.py [configs] [seed]   -- one line per mismatch or rejected configuration, summary at the
end; exit code 1 on any mismatch."""
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
import oracle_lib as oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = rejected = total = 0
reasons = {}
verbose = os.environ.get('FUZZ_VERBOSE')
t0 = time.time()


def content(kind, w, h, seed):
    imgs = synth.stereo_batch(seed, 1, w, h)
    if kind == 1:
        imgs = np.clip(imgs.astype(np.int16) + rng.integers(-12, 13, imgs.shape), 0, 255).astype(np.uint8)
    elif kind == 2:
        imgs = np.clip(imgs.astype(np.int16) // 3 + int(rng.integers(0, 150)), 0, 255).astype(np.uint8)
    elif kind == 3:                                  # saturated / clipped scene
        imgs = np.clip((imgs.astype(np.int16) - 100) * 3, 0, 255).astype(np.uint8)
    elif kind == 4:                                  # structured pattern, shifted for the right image
        y, x = np.mgrid[0:h, 0:w]
        per = int(rng.integers(24, 120))
        t = np.abs(((x + (y if rng.random() < 0.5 else 0)) % per) - per // 2) * (510.0 / per)
        left = np.clip(t, 0, 255).astype(np.uint8)
        imgs = np.stack([left, np.roll(left, -int(rng.integers(1, 12)), axis=1)])
    elif kind == 5:                                  # pure noise
        imgs = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    return np.ascontiguousarray(imgs)


for it in range(N):
    w = int(rng.integers(200, 1400)); h = int(rng.integers(150, min(720, w) + 1))          # landscape, as the reference's octree needs
    p = _lib.default_params()
    p.orb.nfeatures = int(rng.integers(50, 4600))
    p.orb.scale_factor = float(rng.choice([1.1, 1.2, 1.25, 1.35, 1.5, 2.0]))
    # mostly level counts whose top level still holds a 30 px cell (>= 62 px); now and then one too many, which must be refused cleanly
    fit = 1
    while fit < 8 and min(w, h) / p.orb.scale_factor ** fit >= 64:
        fit += 1
    p.orb.nlevels = int(rng.integers(1, fit + 1)) if rng.random() < 0.95 else min(8, fit + 1)
    p.orb.ini_th_fast = int(rng.integers(8, 40)); p.orb.min_th_fast = int(rng.integers(2, p.orb.ini_th_fast + 1))
    p.line.lsd_nfeatures = int(rng.choice([0, 20, 100, 300, 500, 800]))
    p.line.min_line_length = float(rng.choice([0.0, 0.025, 0.05, 0.1]))
    p.line.lsd_scale = float(rng.choice([0.5, 0.8, 0.8, 1.0, 1.2, 1.2, 1.5, 2.0]))
    p.line.lsd_sigma_scale = float(rng.choice([0.6, 0.75]))
    p.line.lsd_quant = float(rng.choice([1.0, 2.0, 3.0]))
    p.line.lsd_ang_th = float(rng.choice([15.0, 22.5, 30.0, 22.5, 5.0, 45.0, 70.0, 85.0]))      # (85: beyond the cheap alignment test's range -- the reference-expression kernel)
    p.line.lsd_n_bins = int(rng.choice([256, 512, 1024, 1024, 1024, 2048, 5000]))      # (> 1024: the 64-bit-key capacity path, lsd_wide.hip)
    p.line.conv_seed_order = int(rng.random() < 0.7)       # convention C.9: mostly the std::sort order (the default), sometimes the raster order
    p.line.lsd_refine = int(rng.choice([0, 0, 0, 0, 1, 1, 2]))   # LSD_REFINE_STD / ADV on some of the draws
    p.line.lsd_log_eps = float(rng.choice([0.0, 0.0, 1.0, -1.0]))
    p.line.lsd_density_th = float(rng.choice([0.6, 0.6, 0.7, 0.85]))
    p.stereo.fx, p.stereo.bf = float(rng.uniform(300, 900)), float(rng.uniform(30, 400))
    p.stereo.best_lr_matches = int(rng.integers(0, 2))
    kind = int(rng.integers(0, 6))
    if kind == 5 and w * h > 300 * 1000:
        kind = 1                                     # full-size pure noise exceeds the 65535 corners a level can hold
    desc = (f"#{it} {w}x{h} kind {kind} orb({p.orb.nfeatures},{p.orb.scale_factor:.2f},{p.orb.nlevels},{p.orb.ini_th_fast},{p.orb.min_th_fast}) "
            f"lsd(n={p.line.lsd_nfeatures},len={p.line.min_line_length},refine={p.line.lsd_refine},dens={p.line.lsd_density_th},s={p.line.lsd_scale},sig={p.line.lsd_sigma_scale},q={p.line.lsd_quant},"
            f"a={p.line.lsd_ang_th},bins={p.line.lsd_n_bins})")
    try:
        fe = ola.StereoFrontEnd(p, w, h, max_pairs=1)
    except _lib.OlfError as e:                         # a configuration the library refuses (e.g. a pyramid level too small): must be a clean error
        rejected += 1
        reasons[str(e).split(": ")[-1][:90]] = reasons.get(str(e).split(": ")[-1][:90], 0) + 1
        if verbose:
            print("rejected", desc, "--", str(e)[:100], flush=True)
        continue
    imgs = content(kind, w, h, 5000 + it)
    try:
        g = fe.frames(imgs).pair(0)
    except _lib.OlfError as e:
        rejected += 1
        import re
        m = re.search(r"flags=(\d+)", str(e))
        why = {"8": "LSD region / raw segment / pixel-list capacity (pure noise at a large lsd_scale)"}.get(m.group(1), "flags=" + m.group(1)) if m else str(e)[-70:]
        reasons["at run time: " + why] = reasons.get("at run time: " + why, 0) + 1
        print("rejected at run time", desc, "--", str(e)[:100], flush=True)
        continue
    o = oracle.stereo_points(imgs[0], imgs[1], p, cap=p.orb.nfeatures + 2064)
    cap_l = 20000
    ol, orr = oracle.line_extract(imgs[0], p.line, all_cap=cap_l), oracle.line_extract(imgs[1], p.line, all_cap=cap_l)
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
        print("MISMATCH", desc, ":", ",".join(k for k, v in checks.items() if not v), f"(kps {len(g['mvKeys'])}/{len(o['kpsL'])}, lines {len(g['mvKeys_Line'])}/{len(ol['kls'])})", flush=True)
    del fe
for k, v in sorted(reasons.items(), key=lambda kv: -kv[1]):
    print(f"  refused x{v}: {k}")
print(f"FUZZ {'FAILED' if bad else 'OK'}: {total - bad}/{total} configurations bit-identical, {rejected} rejected with an error, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
