#!/bin/bash
# LiDAR view calculation A/B in one box session: dense-grid view launches (NVBX_LIDAR_VIEW_GRID=1, default) against k_mark_view<Lidar> (=0).
# Usage: tools/gpu_lidar_ab.sh TAG
TAG=${1:-lidar_ab}; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp; R=$PWD
for G in 1 0; do
  NVBX_LIDAR_VIEW_GRID=$G timeout 300 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/$TAG/bench_grid$G.json 2> gpurun_out/$TAG/bench_grid$G.err; echo "grid=$G rc=$?"
  python - <<PY
import json
d = json.load(open("gpurun_out/$TAG/bench_grid$G.json"))
print("grid=$G ms/scan", d["ms_per_step"], "exploring", d["ms_per_step_exploring"], {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()}, d["per_step_counts"])
PY
done
(cd /tmp && NVBX_BENCH_MIN_MS=500 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o stats -- python $R/bench.py --workload lidar --steps 50 --warmup 5 --profile-run --no-cpu-baseline > /dev/null 2> $R/gpurun_out/$TAG/prof.err); echo "rocprof rc=$?"
find gpurun_out/$TAG/stats -name "*kernel_trace.csv" -delete
find gpurun_out/$TAG/stats -name "*kernel_stats.csv" -exec head -8 {} \; | cut -c1-260
