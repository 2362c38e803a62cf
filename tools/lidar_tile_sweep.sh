#!/bin/bash
# Build variants of the LiDAR view-marking tile shape (rays per wavefront x segments per ray) and time the LiDAR workload with each.
# Usage (here): [V="tr,tc,seg ..." PFX=LIDAR|CAM] tools/lidar_tile_sweep.sh build     (on the GPU box): [V=... WL=lidar|camera] tools/lidar_tile_sweep.sh run TAG
cd "$(dirname "$0")/.."
V=${V:-"1,4,16 1,2,32 2,2,16"}
PFX=${PFX:-LIDAR}; WL=${WL:-lidar}; ARGS=${ARGS:---steps 60 --warmup 10}
if [ "$1" = build ]; then
  mkdir -p isaac_ros_nvblox_amd/variants
  for v in $V; do IFS=, read tr tc sg <<< "$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DNVBX_${PFX}_TR=$tr -DNVBX_${PFX}_TC=$tc -DNVBX_${PFX}_SEG=$sg \
      -c isaac_ros_nvblox_amd/csrc/tsdf.hip -o /tmp/tsdf_$tr$tc$sg.o &
  done; wait
  for v in $V; do IFS=, read tr tc sg <<< "$v"
    objs=$(ls isaac_ros_nvblox_amd/csrc/*.o | grep -v tsdf.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o isaac_ros_nvblox_amd/variants/libnvblox_hip_$tr$tc$sg.so $objs /tmp/tsdf_$tr$tc$sg.o
  done
  ls -la isaac_ros_nvblox_amd/variants
else
  TAG=${2:-sweep}; mkdir -p gpurun_out/$TAG
  for v in $V; do IFS=, read tr tc sg <<< "$v"
    NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$tr$tc$sg.so timeout 300 python bench.py --workload $WL $ARGS --no-cpu-baseline > gpurun_out/$TAG/lidar_$tr$tc$sg.json 2>/dev/null
    python - <<PY
import json
j=json.load(open("gpurun_out/$TAG/lidar_$tr$tc$sg.json"))
print("$tr x $tc rays x $sg seg:", j["ms_per_step"], {k: round(v["avg_us"],1) for k,v in j["kernels"].items()})
PY
  done
fi
