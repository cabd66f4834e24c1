#!/bin/bash
# seed-sort kernel time per variant and batch size: bash tools/ss_sweep.sh "0 3 4" "64 1536 6144"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for n in $2; do for m in $1; do
  rm -rf /tmp/ssw; OLF_SS_MW=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ssw -o run -- python $R/tools/prof_lines_so.py $n > /tmp/ssw.log 2>&1
  python - "$n" "$m" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/ssw/*kernel_stats.csv')
if not f: print("images", sys.argv[1], "mode", sys.argv[2], "no stats"); sys.exit()
for r in csv.DictReader(open(f[0])):
    if "seedsort" in r["Name"]:
        print("images %5s mode %s  %-40s calls %s avg %9.3f ms" % (sys.argv[1], sys.argv[2], r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done; done
