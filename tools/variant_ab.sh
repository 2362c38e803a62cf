#!/bin/bash
# A/B timing of library variants (isaac_ros_nvblox_amd/variants/libnvblox_hip_<name>.so) within ONE GPU-box session -- boxes differ by
# a few per cent in clocks, so variants are only comparable inside one call.  Usage: tools/variant_ab.sh TAG WORKLOAD "bench args" v0 v1 ...
cd "$(dirname "$0")/.."
TAG=$1; WL=$2; ARGS=$3; shift 3
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for v in "$@"; do
  L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so; [ $v = current ] && L=$PWD/isaac_ros_nvblox_amd/libnvblox_hip.so
  NVBX_LIB=$L timeout 300 python bench.py --workload $WL $ARGS --no-cpu-baseline > gpurun_out/$TAG/${WL}_$v.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("gpurun_out/$TAG/${WL}_$v.json"))
print("$v:", j["ms_per_step"], {k.split("<")[0].replace("void ",""): round(x["avg_us"],1) for k,x in j["kernels"].items()})
PY
done; done
