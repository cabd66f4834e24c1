#!/bin/bash
# Build a variant of liborbline_hip.so next to the production one (build/variants/NAME.so; it travels to the GPU box, select it with OLF_LIB_PATH):
#   bash tools/build_variant.sh NAME "-DOLF_STATS" lsd.hip [more sources recompiled with the flags]
# The production objects are not touched: the listed sources are compiled into build/variants/obj_NAME/ and linked with the other production objects.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/orb_line_slam_amd/csrc; V=$R/build/variants
NAME=$1; FLAGS=$2; shift 2
make -s -C $C -j8
mkdir -p $V/obj_$NAME
OBJS=""
for o in $C/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "${f%.*}" = "$b" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in "$@"; do
  extra=""; [ "$f" = "match.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra $FLAGS -x hip -c $C/$f -o $V/obj_$NAME/${f%.*}.o
  OBJS="$OBJS $V/obj_$NAME/${f%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$NAME.so $OBJS
echo "built $V/$NAME.so"
