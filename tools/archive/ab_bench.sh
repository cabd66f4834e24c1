#!/bin/bash
# A/B of an environment switch on the headline step: bash tools/ab_bench.sh VAR "v1 v2 ..." [bench args]
V=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $V=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$V=$v', 'ms_per_step', d['ms_per_step'], 'front', s.get('lsd_front'), 'grow', s.get('lsd_grow'), 'pyr', s.get('orb_pyramid'), 'fast', s.get('orb_fast_cells'))"
done; done
