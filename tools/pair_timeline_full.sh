#!/bin/bash
# every kernel and copy of ONE single-pair olf_stereo_frames call (the last of a few), with gaps: bash tools/pair_timeline_full.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
cat > /tmp/pair1.py <<PY
import sys; sys.path.insert(0, "$R")
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
fe = ola.StereoFrontEnd(_lib.default_params(), 1242, 375, max_pairs=1)
imgs = synth.stereo_batch(11, 1, 1242, 375)
for _ in range(4): fe.frames(imgs)
PY
rm -rf /tmp/pt; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pt -o run -- python /tmp/pair1.py > /tmp/pt.log 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("/tmp/pt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("olf::", "").split("(")[0][:40], "q" + r.get("Queue_Id", "?")))
for f in glob.glob("/tmp/pt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + str(r.get("Bytes", r.get("Size", "0")) or 0), "cp"))
rows.sort()
# the last call: from the last host-to-device copy of the images (the largest H2D) on
def nbytes(r):
    try: return int(r[2].split()[-1])
    except ValueError: return 0
h2d = [i for i, r in enumerate(rows) if r[2].startswith("COPY") and "HOST_TO_DEVICE" in r[2].upper() and (nbytes(r) > 500000 or nbytes(r) == 0)]
start = h2d[-1]
t0 = rows[start][0]; prev = t0
for s, e, k, q in rows[start:]:
    print("%8.3f %8.3f %7.3f  gap %6.3f  %-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, (s - prev) / 1e6, q, k))
    prev = max(prev, e)
PY
