#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4n; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -q -x -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
for rep in 1 2; do
OLF_ONE_STREAM=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream new"
OLF_UPGRAD=0 OLF_ONE_STREAM=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream UPGRAD=0"
timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream new"
OLF_SCHED=18 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream SCHED=18"
OLF_SCHED=18 OLF_FAST_NT=128 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream SCHED=18 NT=128"
OLF_SCHED=0 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream SCHED=0"
done | tee $O/stages.txt
