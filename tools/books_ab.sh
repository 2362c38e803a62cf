#!/bin/bash
# A/B of two library builds on the room, the hall and the 8-camera batch in ONE box session.   usage: tools/books_ab.sh TAG v0 v1
cd "$(dirname "$0")/.."
TAG=$1; shift; mkdir -p gpurun_out/$TAG
for rep in 1 2; do for v in "$@"; do
  L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so; [ $v = current ] && L=$PWD/isaac_ros_nvblox_amd/libnvblox_hip.so
  NVBX_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-parity > gpurun_out/$TAG/room_$v.json 2>/dev/null
  NVBX_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/$TAG/k20_$v.json 2>/dev/null
  NVBX_LIB=$L timeout 300 python bench.py --scene hall --no-cpu-baseline --no-parity > gpurun_out/$TAG/hall_$v.json 2>/dev/null
  NVBX_LIB=$L timeout 300 python bench.py --workload multicam --cameras 8 --steps 100 --warmup 20 --no-cpu-baseline --no-parity > gpurun_out/$TAG/m8_$v.json 2>/dev/null
  python - <<PY
import json
o = []
for w in ("room", "k20", "hall", "m8"):
    d = json.load(open("gpurun_out/$TAG/%s_$v.json" % w)); o.append("%s %.4f (fused %.1f us)" % (w, d["ms_per_step"], d["kernels"]["k_integrate_tsdf_color"]["avg_us"]))
print("$v:", " | ".join(o))
PY
done; done
