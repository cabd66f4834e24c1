#!/bin/bash
# instruction counts of the growth kernel for several library variants, one GPU call: bash tools/pmc_grow_ab.sh label:ENV=V ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  rm -rf /tmp/pb_$label
  env $envs timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d /tmp/pb_$label -o run -- python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 1 --warmup 0 > /tmp/pb_$label.log 2>&1
  echo "== $label"; python $R/tools/pmc_budget.py /tmp/pb_$label 6144 | grep -E "k_lsd_grow|total"
done
