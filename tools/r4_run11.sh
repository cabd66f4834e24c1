#!/bin/bash
# round 4, run 11: batch pipelining through the input event (olf_ctx_set_input_event), single-step threshold 2, full GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4q; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
stage() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stages_ms_per_step',{})
print('$1', 'fps', d['value'], 'ms_per_step', d['ms_per_step'], {k: round(v,1) for k,v in s.items()})"; }
B="--no-cpu-baseline --no-extras --no-isolated"
for rep in 1 2 3; do
timeout 300 python $R/bench.py $B --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "pipelined"
timeout 300 python $R/bench.py $B --no-pipeline --steps 8 --warmup 2 2>/dev/null | tail -1 | stage "--no-pipeline"
done | tee $O/stages.txt
timeout 600 python $R/bench.py --no-cpu-baseline --steps 8 --warmup 2 2>$O/bench_err.txt | tail -1 > $O/bench_C3.json
python - <<PY | tee -a $O/stages.txt
import json
d=json.load(open("$O/bench_C3.json"))
print(d["value"], d["ms_per_step"], d.get("pcie_inclusive"), d.get("pair_latency_ms"))
PY
