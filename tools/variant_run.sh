#!/bin/bash
# Build the LSD kernels with an extra -D flag, run a command on the GPU box, restore the production library.
#   bash tools/variant_run.sh OLF_STATS 'python tools/prof_stats.py' [tail lines]
R=/root/repo; C=$R/orb_line_slam_amd/csrc
make -s -C $C -j8 2>&1 | grep -E "error" 
mkdir -p /tmp/_prod; cp $C/liborbline_hip.so $C/lsd.o $C/lsd_grow.o /tmp/_prod/
( cd $C && for f in lsd lsd_grow; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$1 -x hip -c $f.hip -o $f.o 2>&1 | grep -E "error"; done; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liborbline_hip.so *.o )
( cd $R && timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$2" 2>&1 | grep -v "^\[gpurun\] send" | tail -${3:-6} )
cp /tmp/_prod/liborbline_hip.so /tmp/_prod/lsd.o /tmp/_prod/lsd_grow.o $C/; touch $C/*.o $C/liborbline_hip.so
