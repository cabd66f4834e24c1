"""Per-kernel VALU / SALU wave-instruction budget of one command: python tools/pmc_budget.py <pmc_dir> <images>"""
import csv, glob, collections, sys
d, n = sys.argv[1], float(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]] += float(r["Counter_Value"])
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:26]:
    print("%-44s VALU/img %7.3fM  SALU/img %7.3fM  share %5.1f%%" % (k, v["SQ_INSTS_VALU"] / n / 1e6, v["SQ_INSTS_SALU"] / n / 1e6, 100 * v["SQ_INSTS_VALU"] / tot))
print("total VALU wave-instructions per image: %.2fM" % (tot / n / 1e6))
