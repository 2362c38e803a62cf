# quick GPU check of a kernel change: the colour / pipeline / batch / sequence tests, then the camera lines (default, zero-copy) and the decay line
TAG=${1:-rq2}; mkdir -p gpurun_out/$TAG
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_gpu_round4.py tests/test_gpu_sequences.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.log
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "camera rc=$?"
timeout 400 python bench.py --no-cpu-baseline --zero-copy-deferral > gpurun_out/$TAG/bench_zc.json 2> gpurun_out/$TAG/bench_zc.err; echo "camera zc rc=$?"
timeout 400 python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline > gpurun_out/$TAG/bench_decay.json 2> gpurun_out/$TAG/bench_decay.err; echo "decay rc=$?"
timeout 400 python bench.py --workload multicam --cameras 8 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/$TAG/bench_multicam8.json 2> gpurun_out/$TAG/bench_multicam8.err; echo "multicam8 rc=$?"
python - <<PY
import json
for w in ('','_zc','_decay','_multicam8'):
    try:
        d=json.loads(open('gpurun_out/$TAG/bench%s.json' % w).read().strip().split('\n')[-1])
        print(w or 'camera', d['ms_per_step'], d.get('ms_per_step_revisit'), d.get('ms_per_step_classic_order'), {k:v for k,v in d.get('color_deferral',{}).items() if k.startswith('ms_')}, (d.get('parity') or {}).get('ok'))
        print('   ', {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    except Exception as e: print(w, 'ERR', e)
PY
