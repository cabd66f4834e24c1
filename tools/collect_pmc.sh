#!/bin/bash
# only the counter passes of collect_profiles.sh (the summaries bench.py matches by source hash): bash tools/collect_pmc.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-snap}
O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
BENCH1="python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 1 --warmup 0"
rm -rf /tmp/pf /tmp/pw /tmp/pb
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pf -o run -- $BENCH1 > /tmp/pf.log 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d /tmp/pw -o run -- $BENCH1 > /tmp/pw.log 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw 6144 $O/${T}_pmc_hbm_traffic > $O/${T}_pmc_hbm_traffic_step6144.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d /tmp/pb -o run -- $BENCH1 > /tmp/pb.log 2>&1
python $R/tools/pmc_budget.py /tmp/pb 6144 $O/${T}_valu_budget.json > $O/${T}_valu_budget_per_kernel.txt 2>&1
ls -la $O
