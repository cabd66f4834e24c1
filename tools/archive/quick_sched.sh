#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for sc in 1 2 3 4 0; do echo "OLF_SCHED=$sc"; OLF_SCHED=$sc python $R/bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-200; done
