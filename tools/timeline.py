"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV: python tools/timeline.py <dir>
Prints, for the last step (from its first k_ingest on), start / end / duration (ms) and queue of every kernel longer than 0.3 ms."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?")))
rows.sort()
ing = [i for i, r in enumerate(rows) if "ingest" in r[2]]
first = ing[-1] if ing else 0
while first - 1 in set(ing): first -= 1
t0 = rows[first][0]
sel = rows[first:]
qs = sorted({r[3] for r in sel})
print("queues:", qs, "kernels:", len(sel), "span ms: %.2f" % ((max(r[1] for r in sel) - t0) / 1e6))
for s, e, k, q in sel:
    if (e - s) > int(__import__("os").environ.get("TL_MIN_NS", "300000")):
        print("%8.2f %8.2f %8.2f  q%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, qs.index(q), k))
