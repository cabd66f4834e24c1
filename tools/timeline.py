"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV: python tools/timeline.py <dir>
Prints, for the last step (from its first k_ingest on), start / end / duration (ms) and queue of every kernel longer than 0.3 ms."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?")))
rows.sort()
# the last COMPLETE step on both queues: a step ends with the batched SearchByBoW (bench.py runs it behind the fused entry), so the step is everything
# that started after the previous step's k_search_by_bow ended -- the line stream's kernels start before the ORB stream's k_ingest, which is why the
# ingest-anchored version of round 3 showed the ORB queue only
# (with the deferred join the step's last kernel is the line matcher's k_ratio_mutual, behind the point matcher's k_search_by_bow)
last = [i for i, r in enumerate(rows) if "k_ratio_mutual" in r[2]] or [i for i, r in enumerate(rows) if "k_search_by_bow" in r[2]]
bow = last
if len(bow) >= 2:
    t_prev_end, t_last_end = rows[bow[-2]][1], rows[bow[-1]][1]
    sel = [r for r in rows if r[0] >= t_prev_end and r[0] <= t_last_end]
else:
    ing = [i for i, r in enumerate(rows) if "ingest" in r[2]]
    first = ing[-1] if ing else 0
    while first - 1 in set(ing): first -= 1
    sel = rows[first:]
t0 = sel[0][0]
qs = sorted({r[3] for r in sel})
print("queues:", qs, "kernels:", len(sel), "span ms: %.2f" % ((max(r[1] for r in sel) - t0) / 1e6))
for s, e, k, q in sel:
    if (e - s) > int(__import__("os").environ.get("TL_MIN_NS", "300000")):
        print("%8.2f %8.2f %8.2f  q%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, qs.index(q), k))
