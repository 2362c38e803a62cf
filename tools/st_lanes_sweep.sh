#!/bin/bash
# lanes per ray of k_sphere_trace for camera batches (color.hip sphere_trace_lanes): ms per step and the kernel's duration per setting
# Usage: tools/st_lanes_sweep.sh OUTDIR [cameras...]
OUT=${1:-gpurun_out/st_lanes}; shift
CAMS=("$@"); [ ${#CAMS[@]} -eq 0 ] && CAMS=(8 4 2)
mkdir -p $OUT
for C in "${CAMS[@]}"; do
  for L in 8 4 2 1; do
    NVBX_ST_LANES=$L timeout 600 python bench.py --workload multicam --cameras $C --steps 50 --warmup 10 --no-cpu-baseline > $OUT/mc${C}_l$L.json 2> $OUT/mc${C}_l$L.err
    python - <<PY
import json
d = json.load(open("$OUT/mc${C}_l$L.json"))
k = d["kernels"]
print("cameras $C lanes $L: ms_per_step %.4f (revisit %.4f)  k_sphere_trace %.2f us  k_integrate_color %.2f us" % (d["ms_per_step"], d.get("ms_per_step_revisit", 0), k["k_sphere_trace"]["avg_us"], k["k_integrate_color"]["avg_us"]))
PY
  done
done
