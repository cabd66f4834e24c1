#!/bin/bash
# round 4, run 10: schedule 5 as default + LBD gradient images in the sort's shadow (OLF_LBD_PRE) + 16-byte table clear + single-step threshold variants
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1200 python -m pytest $R/tests/test_line_gpu.py $R/tests/test_lsd_grow_gpu.py $R/tests/test_bow_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee $O/pytest.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for rep in 1 2; do
timeout 300 python $R/bench.py $B --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream default(LBD_PRE=1)"
OLF_LBD_PRE=0 timeout 300 python $R/bench.py $B --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream LBD_PRE=0"
for v in single2 single3; do
OLF_LIB_PATH=$R/build/variants/$v.so timeout 300 python $R/bench.py $B --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream $v"
done
done | tee $O/stages.txt
for v in "" single2 single3; do
OLF_LIB_PATH=${v:+$R/build/variants/$v.so} OLF_ONE_STREAM=1 timeout 300 python $R/bench.py $B --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream ${v:-production}"
done | tee -a $O/stages.txt
for v in single2 single3; do
OLF_LIB_PATH=$R/build/variants/$v.so timeout 600 python -m pytest $R/tests/test_lsd_grow_gpu.py $R/tests/test_line_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -1 | tee -a $O/pytest.txt
done
