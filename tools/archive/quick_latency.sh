#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
python -m pytest $R/tests/test_lsd_grow_gpu.py $R/tests/test_line_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -1
python $R/tools/pair_latency.py 2>&1 | tail -6
OLF_ONE_STREAM=1 OLF_SWEEP_NW=4,8,16 python $R/tools/grow_sweep.py 1 8 128 512 1024 2>&1 | tail -5
