#!/bin/bash
# per-kernel times of one one-stream bench step: bash tools/quick_kstats.sh [rows]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks1; OLF_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o run -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 $KS_ARGS > /tmp/ks1.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ks1/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print("%-60s calls %4s avg %9.3f ms  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
