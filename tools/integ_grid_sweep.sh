#!/bin/bash
# Time the LiDAR workload with different grid sizes of k_integrate_tsdf (grid-stride over the view list; NVBX_INTEG_GRID overrides the
# built-in choice).  Run on the GPU box: tools/integ_grid_sweep.sh TAG [grids...]
cd "$(dirname "$0")/.."
TAG=${1:-sweep}; shift; G=${@:-"768 1024 2048 4096 8192 65536"}
mkdir -p gpurun_out/$TAG
for g in $G; do
  NVBX_INTEG_GRID=$g timeout 300 python bench.py --workload lidar --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/$TAG/lidar_grid_$g.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("gpurun_out/$TAG/lidar_grid_$g.json"))
print("grid $g:", j["ms_per_step"], {k.split("<")[0]: round(v["avg_us"],1) for k,v in j["kernels"].items()})
PY
done
