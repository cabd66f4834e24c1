#!/bin/bash
# quick check of an LSD front change on the GPU box: line parity tests + one-stream stage table + two-stream bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
python -m pytest $R/tests/test_line_gpu.py $R/tests/test_lsd_grow_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
OLF_ONE_STREAM=1 python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('one-stream ms/step', d['ms_per_step']); print({k: round(v, 2) for k, v in d['stages_ms_per_step'].items()})"
python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-200
