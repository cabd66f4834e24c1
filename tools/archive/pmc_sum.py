"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name: python tools/pmc_sum.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:50]
        if sub in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]))
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} total {v:.6g}   per dispatch {v/len(n[k]):.6g}")
