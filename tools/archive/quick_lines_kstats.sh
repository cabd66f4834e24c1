cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/ks1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o run -- python $R/tools/prof_lines.py 4096 > /tmp/ks1.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ks1/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %4s avg %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
