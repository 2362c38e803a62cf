#!/bin/bash
# the 14 x 12 x 3 m hall (bench.py --scene hall: ~2 400 blocks in view, 8 x the room's): residency knobs of the fused launch.   usage: tools/hall_sweep.sh TAG
cd "$(dirname "$0")/.."
TAG=${1:-hall}; mkdir -p gpurun_out/$TAG
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --scene hall --no-cpu-baseline --no-parity > gpurun_out/$TAG/$name.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/$name.json")); print("$name:", d["ms_per_step"], d["ms_per_step_revisit"], {k: round(x["avg_us"],1) for k,x in d["kernels"].items() if k in ("k_mark_view","k_integrate_tsdf_color")})
PY
}
V=$PWD/isaac_ros_nvblox_amd/variants
for rep in 1 2; do
run current X=1
run fused8 NVBX_LIB=$V/libnvblox_hip_fused8.so
run grid2048 NVBX_INTEG_GRID=2048 NVBX_COLOR_GRID=2048
run fused8_grid2048 NVBX_LIB=$V/libnvblox_hip_fused8.so NVBX_INTEG_GRID=2048 NVBX_COLOR_GRID=2048
run grid768 NVBX_INTEG_GRID=768 NVBX_COLOR_GRID=768
done
