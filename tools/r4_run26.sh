#!/bin/bash
# round 4, run 26: k_lsd_keys_tiled (2-D tiles, 8 % frame) against the run form (73 % halo rows)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4au; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests/test_line_gpu.py $R/tests/test_lsd_grow_gpu.py $R/tests/test_seedsort_gpu.py $R/tests/test_lsd_refine_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2 | tee $O/pytest.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for v in 1 0; do
OLF_KEYS_TILED=$v OLF_ONE_STREAM=1 timeout 300 python $R/bench.py $B --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream KEYS_TILED=$v"
done | tee $O/stages.txt
for rep in 1 2 3; do for v in 1 0; do
OLF_KEYS_TILED=$v timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "KEYS_TILED=$v"
done; done | tee -a $O/stages.txt
