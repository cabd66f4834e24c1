#!/bin/bash
# round 4, run 12: wave priority (s_setprio) of the ORB-stream guests / of the growth agents
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for rep in 1 2 3; do
timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "production"
for v in guest3 guest1 agent3; do
OLF_LIB_PATH=$R/build/variants/$v.so timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "$v"
done
done | tee $O/stages.txt
