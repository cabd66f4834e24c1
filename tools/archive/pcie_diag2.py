"""python tools/pcie_diag2.py run | show <dir>: two pinned PCIe-leg runs in one process under rocprofv3 --kernel-trace --memory-copy-trace; `show` prints, per run,
the copies' rates by direction and the fused entry's span per batch."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if sys.argv[1] == "run":
    import torch
    from orb_line_slam_amd import _lib
    from orb_line_slam_amd.pipeline import pcie_inclusive_rate
    torch.cuda.set_stream(torch.cuda.Stream())
    for k in range(2):
        r = pcie_inclusive_rate(_lib.default_params(), 1242, 375, pairs=3072, batches=5, producer="pinned")
        print("run", k, r["value"], r["batch_interval_ms"], flush=True)
else:
    import csv, glob
    d = sys.argv[2]
    cp = []
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Bytes", r.get("Size", 0)) or 0)))
    cp.sort()
    big = [c for c in cp if c[3] > 50_000_000]
    print("copies > 50 MB:", len(big))
    t0 = big[0][0] if big else 0
    for s, e, dr, b in big:
        print("%9.1f ms  %-16s %7.1f MB  %7.1f ms  %6.1f GB/s" % ((s - t0) / 1e6, dr, b / 1e6, (e - s) / 1e6, b / max(e - s, 1)))
