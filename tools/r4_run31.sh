#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4bb; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests/test_line_gpu.py $R/tests/test_multirank_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -1 | tee $O/pytest.txt
B="--no-cpu-baseline --no-extras --no-isolated"
for rep in 1 2 3 4; do for v in "" "--no-deferred-join"; do
timeout 300 python $R/bench.py $B $v --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${v:-deferred join}', d['value'], d['ms_per_step'])"
done; done | tee $O/stages.txt
