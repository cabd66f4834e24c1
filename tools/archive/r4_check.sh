#!/bin/bash
# round 4 routine check on the GPU box: the whole -m gpu suite, the one-stream stage table and the two-stream step (twice each), optionally against a
# reference library:  bash tools/r4_check.sh TAG [REF.so]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-chk}; REF=$2; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1500 python -m pytest $R/tests -q -x -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
for rep in 1 2; do
  OLF_ONE_STREAM=1 timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 $BENCH_ARGS 2>/dev/null | tail -1 | stage "one-stream new"
  [ -n "$REF" ] && OLF_ONE_STREAM=1 OLF_LIB_PATH=$R/$REF timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 $BENCH_ARGS 2>/dev/null | tail -1 | stage "one-stream ref"
  timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 $BENCH_ARGS 2>/dev/null | tail -1 | stage "two-stream new"
  [ -n "$REF" ] && OLF_LIB_PATH=$R/$REF timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 $BENCH_ARGS 2>/dev/null | tail -1 | stage "two-stream ref"
done | tee $O/stages.txt
