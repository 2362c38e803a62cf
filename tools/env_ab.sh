# A/B of runtime environment settings inside one GPU session (camera line, driver flags).  usage: tools/env_ab.sh TAG "ENV1=.. ENV2=.." "ENV..." ...
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for rep in 1 2; do i=0; for E in "$@"; do
  i=$((i+1))
  env $E timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/e${i}_$rep.json 2> gpurun_out/$TAG/e${i}_$rep.err
  python - <<PY
import json
j=json.load(open("gpurun_out/$TAG/e${i}_$rep.json"))
print("[$E]", j["ms_per_step"], j.get("ms_per_step_revisit"), j["frame_latency"]["wall_ms"]["p50"], {k: round(x["avg_us"],2) for k,x in j["kernels"].items()})
PY
done; done
