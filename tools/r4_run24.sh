#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4ar; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for rep in 1 2; do
for p in -1 1 0; do
OLF_S2_PRIO=$p timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "line stream priority $p"
done; done | tee $O/stages.txt
for p in -1 1 0; do echo "== OLF_S2_PRIO=$p"; OLF_S2_PRIO=$p python $R/tools/pcie_diag.py 1 2>&1 | grep -v "^/opt"; done | tee $O/pcie.txt
