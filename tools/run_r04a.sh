set -x
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests/test_gpu_round4.py "tests/test_gpu_pipeline.py::test_dynamic_mapping_frame_keeps_the_pipeline" -x -q --durations=12 > gpurun_out/r04a/pytest_new.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r04a/pytest_new.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench_k20.json 2> gpurun_out/r04a/bench_k20.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r04a/bench_k20.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a/bench_k20.json'))
print(json.dumps({k:d[k] for k in ('value','ms_per_step','ms_per_step_revisit','ms_per_step_classic_order','parity','color_deferral')}, indent=1)[:3000])
print({k:(v['avg_us'],v['launches_per_step']) for k,v in d['kernels'].items()})
PY
