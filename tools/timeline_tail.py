"""Kernels of the last `ms` milliseconds of a rocprofv3 --kernel-trace CSV: python tools/timeline_tail.py <dir> [ms] [min_ms]"""
import csv, glob, sys
d = sys.argv[1]; span = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0; mn = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("olf::", "").split("(")[0][:44], r.get("Queue_Id", "?")))
rows.sort()
end = max(r[1] for r in rows)
sel = [r for r in rows if r[0] >= end - span * 1e6]
t0 = sel[0][0]; qs = sorted({r[3] for r in sel})
for s, e, k, q in sel:
    if (e - s) >= mn * 1e6: print("%8.2f %8.2f %8.2f  q%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, qs.index(q), k))
