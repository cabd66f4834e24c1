#!/bin/bash
# two-stream timeline of one bench step: bash tools/quick_timeline.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o run -- python $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /tmp/ks.log 2>&1
python $R/tools/timeline.py /tmp/ks 2>&1 | head -90
