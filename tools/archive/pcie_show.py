import csv, glob, sys
d = sys.argv[1]
k = []
for f in glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        k.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"][12:], "s" + r["Stream_Id"]))
k.sort()
ing = [x for x in k if "k_ingest" in x[2]]
for which in (3, len(ing) - 3):
    t0 = ing[which][0]
    print("==== around ingest #%d" % which)
    for s, e, n, q in k:
        if t0 - 340e6 <= s <= t0 + 120e6 and e - s > 1.5e6:
            print("%8.1f %8.1f %7.1f %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
