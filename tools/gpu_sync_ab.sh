#!/bin/bash
# polled stream wait (NVBX_SPIN_SYNC=1, default) against hipStreamSynchronize (=0) in one box session: node-cadence line and the camera line's frame latency
TAG=${1:-sync_ab}; mkdir -p gpurun_out/$TAG
for S in 1 0 1 0; do
  NVBX_SPIN_SYNC=$S timeout 300 python bench.py --workload node --no-parity > gpurun_out/$TAG/node_$S.json 2>/dev/null
  NVBX_SPIN_SYNC=$S timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/$TAG/cam_$S.json 2>/dev/null
  python - <<PY
import json
n = json.load(open("gpurun_out/$TAG/node_$S.json")); c = json.load(open("gpurun_out/$TAG/cam_$S.json"))
print("spin=$S node ms/s", n["ms_per_simulated_second"], "slice as called", n["tags_as_called"]["esdf/slice"]["ms_per_call"], "| camera k20", c["ms_per_step"], "latency", c["frame_latency"]["wall_ms"], c["frame_latency"]["gpu_ms"]["p50"])
PY
done
