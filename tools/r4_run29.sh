#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4az; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests/test_line_gpu.py $R/tests/test_match_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee $O/pytest.txt
B="--no-cpu-baseline --no-extras --no-isolated"
rm -rf /tmp/ks1; OLF_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o run -- python $R/bench.py $B --steps 3 --warmup 1 > /tmp/ks1.log 2>&1
python - <<PY | tee $O/kstats.txt
import csv, glob
for r in csv.DictReader(open(glob.glob('/tmp/ks1/*kernel_stats.csv')[0])):
    if any(k in r["Name"] for k in ("k_lines_", "k_lbd_desc", "k_stereo_", "k_octree", "k_cells")): print("%-60s calls %4s avg %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
for rep in 1 2 3; do timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('step', d['value'], d['ms_per_step'])"; done | tee -a $O/kstats.txt
