#!/bin/bash
# build one .hip with an extra -D, run a command on the GPU, restore: bash /tmp/vr.sh <file-stem> <DEF> '<cmd>'
R=/root/repo; C=$R/orb_line_slam_amd/csrc
make -s -C $C -j8 2>&1 | grep -E "error"
mkdir -p /tmp/_prod2; cp $C/liborbline_hip.so $C/$1.o /tmp/_prod2/
( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -D$2 -x hip -c $1.hip -o $1.o 2>&1 | grep -E "error"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o liborbline_hip.so *.o )
( cd $R && timeout 1500 /usr/local/graft/bin/gpurun --timeout 600 -- "$3" 2>&1 | grep -v "^\[gpurun\]" | tail -${4:-6} )
cp /tmp/_prod2/liborbline_hip.so /tmp/_prod2/$1.o $C/; touch $C/*.o $C/liborbline_hip.so
