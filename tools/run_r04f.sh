mkdir -p gpurun_out/r04f
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_pipeline.py tests/test_cpp_facade.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r04f/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04f/pytest.log
for a in "" "--separate-front-end"; do
timeout 600 python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline $a > gpurun_out/r04f/bench_decay$a.json 2> gpurun_out/r04f/bench_decay$a.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/r04f/bench_decay$a.json'))
print("$a", {k:d[k] for k in ('value','ms_per_step')}, d['block_stats_ms_per_step'])
print({k:(round(v['avg_us'],1),round(v['launches_per_step'],2)) for k,v in d['kernels'].items()})
PY
done
