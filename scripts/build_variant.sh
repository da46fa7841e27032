#!/bin/bash
# A variant build of the engine for kernel experiments: scripts/build_variant.sh NAME FILE.hip "-DFLAG=..." [FILE2.hip ...]
# recompiles the named sources with the extra flags and links them with the product objects into
# scl_slam_amd/lib/variants/libscl_engine_NAME.so (load it with SCL_ENGINE_LIB=...).  The product library is not touched.
set -e
NAME=$1; shift
FLAGS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/scl_slam_amd/csrc
OUT=$ROOT/scl_slam_amd/lib/variants
mkdir -p $OUT /tmp/scl_variant_$NAME
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-result -I$ROOT/include -I$CSRC"
OBJS=""
for f in engine sc_distance ringkey_topk make_sc icp voxel sharded_front sc_screen sc_masked sc_matrix messages iris device_sort; do
  if echo " $* " | grep -q " $f.hip "; then
    /opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -c $CSRC/$f.hip -o /tmp/scl_variant_$NAME/$f.o
    OBJS="$OBJS /tmp/scl_variant_$NAME/$f.o"
  else
    OBJS="$OBJS $CSRC/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libscl_engine_$NAME.so $OBJS -ldl
echo "built $OUT/libscl_engine_$NAME.so"
