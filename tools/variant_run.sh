#!/bin/bash
# Build lsd.hip with an extra -D flag, run a command on the GPU box, restore the production library.
#   bash tools/variant_run.sh OLF_STATS 'python tools/prof_stats.py'
R=/root/repo; C=$R/orb_line_slam_amd/csrc
cp $C/liborbline_hip.so /tmp/_prod.so; cp $C/lsd.o /tmp/_prod_lsd.o; cp $C/lsd_grow.o /tmp/_prod_lsd_grow.o
( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$1 -x hip -c lsd.hip -o lsd.o 2>&1 | grep -E "error"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$1 -x hip -c lsd_grow.hip -o lsd_grow.o 2>&1 | grep -E "error"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liborbline_hip.so *.o )
( cd $R && timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$2" 2>&1 | grep -v "^\[gpurun\] send" | tail -${3:-6} )
cp /tmp/_prod.so $C/liborbline_hip.so; cp /tmp/_prod_lsd.o $C/lsd.o; cp /tmp/_prod_lsd_grow.o $C/lsd_grow.o; touch $C/lsd.o $C/lsd_grow.o $C/liborbline_hip.so
