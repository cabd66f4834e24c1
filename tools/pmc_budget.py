"""Per-kernel VALU / SALU wave-instruction budget of one command: python tools/pmc_budget.py <pmc_dir> <images> [json_out]
With json_out: the per-image counts per kernel plus the hash of the kernel sources they were collected on (bench.py sets the dominant kernel's vector
instructions against the part's issue rate with it, as it does with the traffic summary)."""
import csv, glob, collections, sys, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
d, n = sys.argv[1], float(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:26]:
    print("%-44s VALU/img %7.3fM  SALU/img %7.3fM  share %5.1f%%" % (k[:44], v["SQ_INSTS_VALU"] / n / 1e6, v["SQ_INSTS_SALU"] / n / 1e6, 100 * v["SQ_INSTS_VALU"] / tot))
print("total VALU wave-instructions per image: %.2fM" % (tot / n / 1e6))
if len(sys.argv) > 3:
    import bench
    json.dump({"source_hash": bench.source_hash(), "images": n, "total_valu_per_image": tot / n,
               "kernels": {k: {"valu_per_image": v["SQ_INSTS_VALU"] / n, "salu_per_image": v["SQ_INSTS_SALU"] / n} for k, v in acc.items()}},
              open(sys.argv[3], "w"), indent=1)
