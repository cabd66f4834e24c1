"""Kernel timeline of ONE stereo pair through the fused entry (the drop-in's online shape): run under rocprofv3 --kernel-trace
     rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o run -- python tools/pair_timeline.py run
     python tools/pair_timeline.py show /tmp/pt
`show` prints every kernel of the last call with start / end / duration in ms relative to the call's first kernel, per queue."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
if sys.argv[1] == "run":
    import orb_line_slam_amd as ola
    from orb_line_slam_amd import synth, _lib
    fe = ola.StereoFrontEnd(_lib.default_params(), 1242, 375, max_pairs=1)
    imgs = synth.stereo_batch(11, 1, 1242, 375)
    for _ in range(4):
        fe.frames(imgs)
else:
    import csv, glob
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "?")))
    rows.sort()
    ing = [i for i, r in enumerate(rows) if "k_ingest" in r[2]]
    # the last call: from the kernel that follows the previous call's last kernel (the line stream starts before the ORB stream's k_ingest)
    last = ing[-1]
    prev_end = max(r[1] for r in rows[:last] if r[0] < rows[ing[-2]][0] + (rows[last][0] - rows[ing[-2]][0]) // 2) if len(ing) > 1 else 0
    sel = [r for r in rows if r[0] >= prev_end]
    t0 = sel[0][0]
    qs = sorted({r[3] for r in sel})
    print("kernels:", len(sel), "span ms: %.3f" % ((max(r[1] for r in sel) - t0) / 1e6))
    for s, e, k, q in sel:
        if e - s > 20000:
            print("%8.3f %8.3f %8.3f  q%d %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, qs.index(q), k))
