#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4aw; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for rep in 1 2 3; do for v in 5 6; do
OLF_SCHED=$v timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "schedule $v"
done; done | tee $O/stages.txt
OLF_SCHED=6 timeout 600 python -m pytest $R/tests/test_line_gpu.py -q -x -m gpu -p no:cacheprovider -k "stereo_frames" 2>&1 | tail -1
