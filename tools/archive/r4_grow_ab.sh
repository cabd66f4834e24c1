#!/bin/bash
# round 4, growth agent A/B: parity of every OLF_GROW_PF variant, event counts, one-stream stage times per variant, two-stream step against the round-3 library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4a; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for pf in 3 1 7 0; do
  echo "== parity OLF_GROW_PF=$pf"; OLF_GROW_PF=$pf timeout 900 python -m pytest $R/tests/test_line_gpu.py $R/tests/test_lsd_grow_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
done | tee $O/parity.txt
for pf in 0 3; do echo "== stats PF=$pf"; OLF_GROW_PF=$pf OLF_LIB_PATH=$R/build/variants/stats.so timeout 300 python $R/tools/prof_stats.py 2>&1 | tail -1; done | tee $O/stats.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
for rep in 1 2; do
  for pf in 0 1 3 7; do OLF_ONE_STREAM=1 OLF_GROW_PF=$pf timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream PF=$pf"; done
  OLF_ONE_STREAM=1 OLF_LIB_PATH=$R/build/variants/base.so timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 4 --warmup 2 2>/dev/null | tail -1 | stage "one-stream r3-lib"
done | tee $O/one_stream.txt
for rep in 1 2; do
  for pf in 0 3 7; do OLF_GROW_PF=$pf timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream PF=$pf"; done
  OLF_LIB_PATH=$R/build/variants/base.so timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-isolated --steps 6 --warmup 2 2>/dev/null | tail -1 | stage "two-stream r3-lib"
done | tee $O/two_stream.txt
