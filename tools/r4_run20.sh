#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4ag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for rep in 1 2 3; do
for v in "" brows8 brows16; do
if [ -n "$v" ]; then export OLF_LIB_PATH=$R/build/variants/$v.so; else unset OLF_LIB_PATH; fi
timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "border rows ${v:-4 (production)}"
done; done | tee $O/stages.txt
