#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
$R/tools/micro/issue_mix 2>&1 | tee $O/issue_mix.txt
timeout 1500 python -m pytest $R/tests -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
OLF_ONE_STREAM=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream" | tee -a $O/stages.txt
for p in 3072 2560 2048 3584; do
  timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 --pairs $p 2>/dev/null | tail -1 | stage "two-stream pairs=$p" | tee -a $O/stages.txt
done
