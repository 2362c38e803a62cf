cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
for rep in 1 2; do for mr in 256 0 512 1024; do
  E=""; [ $mr != 0 ] && E="NVBX_MARK_RIDERS=$mr"
  env $E timeout 300 python bench.py --scene hall --no-cpu-baseline --no-parity > gpurun_out/r06m/hall_$mr.json 2>/dev/null
  env $E timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/r06m/k20_$mr.json 2>/dev/null
  python - <<PY
import json
h=json.load(open("gpurun_out/r06m/hall_$mr.json")); r=json.load(open("gpurun_out/r06m/k20_$mr.json"))
print("riders $mr (0 = adaptive): hall", h["ms_per_step"], round(h["kernels"]["k_mark_view"]["avg_us"],1), "| room k20", r["ms_per_step"], round(r["kernels"]["k_mark_view"]["avg_us"],1))
PY
done; done
