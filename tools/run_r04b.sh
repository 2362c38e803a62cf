mkdir -p gpurun_out/r04b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04b/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04b/pytest.log
NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_wgt.so python tools/wg_timeline.py 2>/dev/null > gpurun_out/r04b/wgt.json; echo "wgt rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04b/bench_k20.json 2> gpurun_out/r04b/bench_k20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b/bench_k20.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_revisit','ms_per_step_classic_order')}, d['parity']['ok'])
print({k:(v['avg_us'],v['launches_per_step']) for k,v in d['kernels'].items()})
PY
