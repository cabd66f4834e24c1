#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4ak; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_orb_gpu.py $R/tests/test_line_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -1
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for v in "" prev; do
if [ -n "$v" ]; then export OLF_LIB_PATH=$R/build/variants/$v.so; else unset OLF_LIB_PATH; fi
OLF_ONE_STREAM=1 timeout 300 python $R/bench.py $B --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream ${v:-new}"
done | tee $O/stages.txt
for rep in 1 2 3; do
for v in "" prev; do
if [ -n "$v" ]; then export OLF_LIB_PATH=$R/build/variants/$v.so; else unset OLF_LIB_PATH; fi
timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "${v:-new}"
done; done | tee -a $O/stages.txt
