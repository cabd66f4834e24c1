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
         "# FETCH_SIZE on gfx950 under-reports wide coalesced 16 B/lane streams by 2x (MI355X_MICROARCH.md, HBM section); the kernels here issue 1-4 byte",
         "# loads / gathers, for which the counter is uncalibrated, so the raw value is reported (k_sep7's WRITE_SIZE equals pitch x rows exactly).",
         "kernel, launches, FETCH KiB/launch, WRITE KiB/launch, (FETCH+WRITE) bytes per image"]
for k in sorted(F, key=lambda k: -(F[k][0] + W.get(k, (0, 0))[0])):
    f, w = F[k][0], W.get(k, (0.0, 0))[0]
    kern[k] = {"fetch_kib_per_launch": f, "write_kib_per_launch": w, "bytes_per_image": (f + w) * 1024 / n_img}
    lines.append(f"{k}, {F[k][1]}, {f:.0f}, {w:.0f}, {(f + w) * 1024 / n_img:.0f}")
json.dump({"images_per_launch": n_img, "source_hash": source_hash(), "kernels": kern}, open(out + ".json", "w"), indent=1)
open(out + ".txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:12]))
