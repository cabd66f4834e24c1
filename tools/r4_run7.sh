#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4l; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
OLF_FAST_NT=128 timeout 900 python -m pytest $R/tests/test_orb_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee $O/pytest_nt128.txt
timeout 900 python -m pytest $R/tests/test_orb_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee $O/pytest_nt256.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
for rep in 1 2; do for nt in 256 128; do
OLF_FAST_NT=$nt OLF_ONE_STREAM=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream NT=$nt"
OLF_FAST_NT=$nt timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream NT=$nt"
done; done | tee $O/stages.txt
