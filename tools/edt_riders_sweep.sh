cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "component or dynamic or mask" 2>&1 | tail -2
for r in 256 512 768 1024 256 512; do
  NVBX_EDT_RIDERS=$r timeout 300 python bench.py --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("/tmp/b.json"))
print("riders $r:", j["ms_per_step"], {k.split("<")[0].replace("void ",""): round(x["avg_us"],1) for k,x in j["kernels"].items()})
PY
done
python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline | cut -c1-250
