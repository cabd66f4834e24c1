#!/bin/bash
# kernel durations of single-pair olf_stereo_frames calls by workgroups per image: bash tools/kstats_pair.sh "1 2 4"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for G in $1; do
cat > /tmp/pair1.py <<PY
import sys; sys.path.insert(0, "$R")
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
fe = ola.StereoFrontEnd(_lib.default_params(), 1242, 375, max_pairs=1)
_lib.check(_lib.lib().olf_debug_lsd_groups(fe.ctx.handle, $G), "g")
imgs = synth.stereo_batch(11, 1, 1242, 375)
for _ in range(6): fe.frames(imgs)
PY
rm -rf /tmp/pt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o run -- python /tmp/pair1.py > /tmp/pt.log 2>&1
echo "== groups $G"; f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-6
done
