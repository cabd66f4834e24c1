#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4i; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -q -x -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
for v in "1 1" "0 0" "1 0" "0 1"; do set -- $v
  OLF_ONE_STREAM=1 OLF_SEP7=$1 OLF_RESIZE=$2 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream SEP7=$1 RESIZE=$2"
done | tee $O/ab.txt
for v in "1 1" "0 0" "1 1" "0 0"; do set -- $v
  OLF_SEP7=$1 OLF_RESIZE=$2 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream SEP7=$1 RESIZE=$2"
done | tee -a $O/ab.txt
python $R/tools/adv_timing.py 2 64 2048 4096 2>&1 | grep refine | tee $O/adv_production.txt
