# round-4 check of the LDS-local connected components + batched block allocation: the tests that allocate / label, then the three bench lines
TAG=${1:-r04n}; mkdir -p gpurun_out/$TAG
T="tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_lidar.py tests/test_gpu_pipeline.py tests/test_gpu_sequences.py tests/test_gpu_edge_cases.py tests/test_gpu_batch.py tests/test_gpu_multi.py"
timeout 1500 python -m pytest $T -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
timeout 400 python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline > gpurun_out/$TAG/bench_decay.json 2> gpurun_out/$TAG/bench_decay.err; echo "decay rc=$?"
timeout 400 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/$TAG/bench_lidar.json 2> gpurun_out/$TAG/bench_lidar.err; echo "lidar rc=$?"
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "camera rc=$?"
python - <<PY
import json
for w in ('decay','lidar',''):
    try:
        d=json.loads(open('gpurun_out/$TAG/bench%s.json' % ('_'+w if w else '')).read().strip().split('\n')[-1])
        print(w or 'camera', d['ms_per_step'], d.get('ms_per_step_exploring'), d.get('ms_per_step_revisit'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    except Exception as e: print(w, 'ERR', e)
PY
