#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4at; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_bow_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -1
B="--no-cpu-baseline --no-extras --no-isolated"
for v in "" bmw8 bmw4; do
if [ -n "$v" ]; then export OLF_LIB_PATH=$R/build/variants/$v.so; else unset OLF_LIB_PATH; fi
rm -rf /tmp/ks1; OLF_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o run -- python $R/bench.py $B --steps 3 --warmup 1 > /tmp/ks1.log 2>&1
python - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob('/tmp/ks1/*kernel_stats.csv')[0])):
    if "search_by_bow" in r["Name"]: print("${v:-16 waves (production)}", r["Name"][:30], "avg ms", float(r["AverageNs"]) / 1e6)
PY
done | tee $O/bow.txt
unset OLF_LIB_PATH
for rep in 1 2 3; do for v in "" bmw4; do
if [ -n "$v" ]; then export OLF_LIB_PATH=$R/build/variants/$v.so; else unset OLF_LIB_PATH; fi
timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-16 waves}', d['value'], d['ms_per_step'])"
done; done | tee -a $O/bow.txt
