#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; T=r4d; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pb
OLF_LSD_NW=0 OLF_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d /tmp/pb -o run -- python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --pairs 512 --steps 1 --warmup 0 > /tmp/pb.log 2>&1
python $R/tools/pmc_budget.py /tmp/pb 1024 $O/${T}_valu_budget.json > $O/${T}_valu_budget_per_kernel.txt 2>&1; head -30 $O/${T}_valu_budget_per_kernel.txt
rm -rf /tmp/ks1; OLF_ONE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o run -- python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 3 --warmup 1 > /tmp/ks1.log 2>&1
cp $(ls /tmp/ks1/*kernel_stats.csv | head -1) $O/${T}_bench_C3_one_stream_kernel_stats.csv
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/${T}_bench_C3_one_stream_kernel_stats.csv")))[:24]:
    print("%-60s calls %4s avg %9.3f ms  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
timeout 900 python $R/bench.py --no-cpu-baseline 2>$O/bench_err.txt | tail -1 > $O/${T}_bench_C3.json; python -c "
import json; d=json.load(open('$O/${T}_bench_C3.json')); print(d['value'], d['ms_per_step'], d.get('pcie_inclusive'), d.get('pair_latency_ms'), d['roofline'].get('valu_issue_step'))"
tail -3 $O/bench_err.txt
timeout 600 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --sequence 6 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sequence 6:', d['value'], d['ms_per_step'], d['config'].get('search_by_bow_mean_matches'), d['config']['order'])"
