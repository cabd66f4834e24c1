#!/bin/bash
# kernel timeline of one single-pair olf_stereo_frames call (the drop-in's online shape): bash tools/pair_timeline.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
cat > /tmp/pair1.py <<PY
import sys; sys.path.insert(0, "$R")
import orb_line_slam_amd as ola
from orb_line_slam_amd import synth, _lib
fe = ola.StereoFrontEnd(_lib.default_params(), 1242, 375, max_pairs=1)
imgs = synth.stereo_batch(11, 1, 1242, 375)
for _ in range(3): fe.frames(imgs)
PY
rm -rf /tmp/pt; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pt -o run -- python /tmp/pair1.py > /tmp/pt.log 2>&1
python $R/tools/timeline_tail.py /tmp/pt ${1:-14} ${2:-0.02}
