"""Randomised parity fuzz of the brute-force matchers (knn2 / match / matchNNR) against the CPU oracle: random set sizes (0 .. 7000, so both
key formats of the MFMA kernel and every tile edge), near-duplicate rows (ties), random ratios.  python tools/fuzz_match.py [cases] [seed]"""
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from orb_line_slam_amd import matcher
import oracle_lib as oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()


def descs(n, base=None):
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if base is not None and n and len(base):             # rows that are a few bit flips away from rows of the other set: real matches
        k = rng.integers(0, n, n // 2)
        d[k] = base[rng.integers(0, len(base), len(k))]
        flips = rng.integers(0, 40, len(k))
        for i, f in zip(k, flips):
            for _ in range(int(f)):
                d[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    for _ in range(n // 10):                              # exact duplicates inside the set: ties
        i, j = rng.integers(0, n, 2)
        d[i] = d[j]
    return d


for it in range(N):
    big = rng.random() < 0.2
    n1 = int(rng.integers(0, 7000 if big else 600)); n2 = int(rng.integers(0, 7000 if big else 600))
    if rng.random() < 0.15:
        n2 = int(rng.choice([0, 1, 2, 31, 32, 33, 63, 64, 65, 4095, 4096, 4097]))
    d2 = descs(n2)
    d1 = descs(n1, d2)
    i0, a, b = matcher.knn2(d1, d2)
    oi, oa, ob = oracle.knn2(d1, d2)
    ok = np.array_equal(i0, oi) and np.array_equal(a, oa) and np.array_equal(b, ob)
    nnr, lr = float(rng.choice([0.6, 0.75, 0.9, 1.0])), bool(rng.integers(0, 2))
    _, m = matcher.match(d1, d2, nnr, best_lr_matches=lr)
    ok = ok and np.array_equal(m, oracle.match_bf(d1, d2, nnr, lr))
    if not ok:
        bad += 1
        print(f"MISMATCH #{it}: n1={n1} n2={n2} nnr={nnr} lr={lr}", flush=True)
print(f"MATCH FUZZ {'FAILED' if bad else 'OK'}: {N - bad}/{N} cases bit-identical, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
