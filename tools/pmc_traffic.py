"""FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, KiB) -> per-kernel HBM traffic summary.
python tools/pmc_traffic.py <fetch_dir> <write_dir> <images_per_launch> <out_prefix>"""
import csv, glob, json, sys, collections, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import source_hash
fd, wd, n_img, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def load(d, counter):
    acc = collections.defaultdict(float); disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0][:60]
                acc[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    return {k: (acc[k] / len(disp[k]), len(disp[k])) for k in acc}


F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
kern = {}
lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and, in a separate pass, --pmc WRITE_SIZE (KiB counters), %d images 1242x375 per launch" % n_img,
         "# Calibration (tools/micro/fetch_calib.hip, profiles/r3_fetch_calibration.txt): FETCH_SIZE counts every memory-side read request at 64 B --",
         "# contiguous reads of 16, 4 or 1 byte per lane all come out at exactly 1/2 of their bytes (128-B requests), a 4-byte gather with every access",
         "# in its own line at 64 B per access; WRITE_SIZE is exact for contiguous stores and 32 B per scattered 4-byte store.  Calibrated bytes =",
         "# 2 x FETCH + WRITE: exact for streaming kernels, an upper bound (128 B per gathered word) for the gather-dominated growth kernels.",
         "kernel, launches, FETCH KiB/launch, WRITE KiB/launch, calibrated (2*FETCH+WRITE) bytes per image, raw (FETCH+WRITE) bytes per image"]
for k in sorted(F, key=lambda k: -(F[k][0] + W.get(k, (0, 0))[0])):
    f, w = F[k][0], W.get(k, (0.0, 0))[0]
    # (bytes_per_image: one launch; bytes_per_image_step: all launches of the profiled command -- one step of bench.py -- e.g. the eight level launches of FAST)
    kern[k] = {"fetch_kib_per_launch": f, "write_kib_per_launch": w, "bytes_per_image": (2 * f + w) * 1024 / n_img, "raw_bytes_per_image": (f + w) * 1024 / n_img,
               "launches": F[k][1], "bytes_per_image_step": (2 * f + w) * 1024 / n_img * F[k][1]}
    lines.append(f"{k}, {F[k][1]}, {f:.0f}, {w:.0f}, {(2 * f + w) * 1024 / n_img:.0f}, {(f + w) * 1024 / n_img:.0f}")
json.dump({"images_per_launch": n_img, "source_hash": source_hash(), "calibration": "bytes_per_image = (2 * FETCH_SIZE + WRITE_SIZE) / images: FETCH_SIZE tallies 128-B read requests at 64 B (profiles/r3_fetch_calibration.txt)", "kernels": kern}, open(out + ".json", "w"), indent=1)
open(out + ".txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
