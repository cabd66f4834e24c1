#!/bin/bash
# A/B of the bench step under environment settings, one GPU call: bash tools/ab_bench.sh "<label>:<ENV=V ...>" ...   (label "base:" = no setting)
# prints per setting: stereo frames/s, ms per step, growth kernel in the step / alone, seed sort front alone
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
STEPS=${AB_STEPS:-4}
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  line=$(env $envs timeout 600 python $R/bench.py --no-cpu-baseline --no-extras --steps $STEPS --warmup 1 ${AB_ARGS:-} 2>/dev/null | grep '^{' | tail -1)
  python - "$label" "$line" <<'PY'
import sys, json
d = json.loads(sys.argv[2]); r = d["roofline"]; st = r.get("stages", {})
print("%-14s %8.1f frames/s  %7.2f ms/step  grow in step %6.2f alone %6.2f  front alone %6.2f  fast alone %5.2f in step %5.2f  rect %5.2f" % (
    sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_ms"], st.get("lsd_grow", {}).get("ms_alone", -1), st.get("lsd_front", {}).get("ms_alone", -1),
    st.get("orb_fast_cells", {}).get("ms_alone", -1), d["stages_ms_per_step"].get("orb_fast_cells", -1), d["stages_ms_per_step"].get("lsd_rect", -1)))
PY
done
