#!/bin/bash
# Build a tuning / instrumentation variant of libnvblox_hip.so: tools/build_variant.sh NAME "-DFLAG=.. -DOTHER"  ->
# isaac_ros_nvblox_amd/variants/libnvblox_hip_NAME.so (select it with NVBX_LIB=...; tools/variant_ab.sh times variants inside one GPU session).
set -e
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=$2
B=build/variant_$NAME; mkdir -p $B isaac_ros_nvblox_amd/variants
C=isaac_ros_nvblox_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result $EXTRA"
PIDS=()
for f in mapper tsdf esdf color mesh maintenance convert esdf3d dynamics ground frames; do
  rm -f $B/$f.o
  ( /opt/rocm/bin/hipcc $FLAGS -c $C/$f.hip -o $B/$f.o ) &
  PIDS+=($!)
done
for p in "${PIDS[@]}"; do wait $p || { echo "variant $NAME: a compile failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o isaac_ros_nvblox_amd/variants/libnvblox_hip_$NAME.so $B/*.o
echo built isaac_ros_nvblox_amd/variants/libnvblox_hip_$NAME.so
