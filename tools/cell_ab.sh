#!/bin/bash
# locality-preserving hash cells (-DNVBX_HASH_CELL=c): times of room / hall / LiDAR / dynamic mapping and FETCH_SIZE of the room's launches, variants in ONE box session.   usage: tools/cell_ab.sh TAG v0 v1 ...
cd "$(dirname "$0")/.."
TAG=$1; shift; mkdir -p gpurun_out/$TAG
for rep in 1 2; do for v in "$@"; do
  L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so; [ $v = current ] && L=$PWD/isaac_ros_nvblox_amd/libnvblox_hip.so
  NVBX_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-parity > gpurun_out/$TAG/room_$v.json 2>/dev/null
  NVBX_LIB=$L timeout 300 python bench.py --scene hall --no-cpu-baseline --no-parity > gpurun_out/$TAG/hall_$v.json 2>/dev/null
  NVBX_LIB=$L timeout 300 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline --no-parity > gpurun_out/$TAG/lidar_$v.json 2>/dev/null
  NVBX_LIB=$L timeout 300 python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline --no-parity > gpurun_out/$TAG/decay_$v.json 2>/dev/null
  python - <<PY
import json
o = []
for w in ("room", "hall", "lidar", "decay"):
    d = json.load(open("gpurun_out/$TAG/%s_$v.json" % w)); k = d["kernels"]
    o.append("%s %.4f" % (w, d["ms_per_step"]) + ((" (mark %.1f fused %.1f mesh %.1f)" % (k["k_mark_view"]["avg_us"], k["k_integrate_tsdf_color"]["avg_us"], k["k_mesh"]["avg_us"])) if w in ("room", "hall") else "") + ((" (resolve %.1f)" % k["k_resolve_view"]["avg_us"]) if w == "lidar" else ""))
print("$v:", " | ".join(o))
PY
done; done
for v in "$@"; do
  L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so; [ $v = current ] && L=$PWD/isaac_ros_nvblox_amd/libnvblox_hip.so
  echo "== FETCH_SIZE KiB per launch, room, $v"
  NVBX_LIB=$L NVBX_BENCH_MIN_MS=30 PMC_TIMEOUT=240 BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --profile-run --with-mesh" bash tools/gpu_pmc.sh $TAG/pmc_$v FETCH_SIZE | grep "k_mark_view\|k_integrate_tsdf_color\|k_mesh"
  echo "== FETCH_SIZE KiB per launch, LiDAR, $v"
  NVBX_LIB=$L NVBX_BENCH_MIN_MS=200 PMC_TIMEOUT=240 BENCH_ARGS="--workload lidar --steps 50 --warmup 5 --no-cpu-baseline --no-parity --profile-run" bash tools/gpu_pmc.sh $TAG/pmcl_$v FETCH_SIZE | grep "k_resolve_view\|k_lidar_sparse\|k_integrate_tsdf"
done
find gpurun_out/$TAG -name "*counter_collection.csv" -size +5M -delete
