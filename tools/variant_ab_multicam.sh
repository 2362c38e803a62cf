#!/bin/bash
# A/B of library variants on the 8-camera batch AND the single-camera line in one box session: tools/variant_ab_multicam.sh TAG v0 v1 ...
cd "$(dirname "$0")/.."
TAG=$1; shift; mkdir -p gpurun_out/$TAG
for v in "$@"; do
  L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so; [ $v = current ] && L=$PWD/isaac_ros_nvblox_amd/libnvblox_hip.so
  NVBX_LIB=$L timeout 300 python bench.py --workload multicam --cameras 8 --steps 100 --warmup 20 --no-cpu-baseline --no-parity > gpurun_out/$TAG/m8_$v.json 2>/dev/null
  NVBX_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/$TAG/cam_$v.json 2>/dev/null
  python - <<PY
import json
m = json.load(open("gpurun_out/$TAG/m8_$v.json")); c = json.load(open("gpurun_out/$TAG/cam_$v.json"))
print("$v: 8 cameras", m["ms_per_step"], {k: round(x["avg_us"], 1) for k, x in m["kernels"].items() if k in ("k_mark_view", "k_integrate_tsdf_color")},
      "| camera k20", c["ms_per_step"], {k: round(x["avg_us"], 1) for k, x in c["kernels"].items() if k in ("k_mark_view", "k_integrate_tsdf_color")})
PY
done
