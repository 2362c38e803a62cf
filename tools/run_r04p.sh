TAG=${1:-r04p}; mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "own_stream or dynamic_front_end" > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.log
timeout 400 python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline > gpurun_out/$TAG/bench_decay.json 2> gpurun_out/$TAG/bench_decay.err; echo "decay rc=$?"
timeout 400 python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline --shared-stream > gpurun_out/$TAG/bench_decay_shared.json 2> gpurun_out/$TAG/bench_decay_shared.err; echo "decay shared rc=$?"
python - <<PY
import json
for w in ('decay','decay_shared'):
    try:
        d=json.loads(open('gpurun_out/$TAG/bench_%s.json' % w).read().strip().split('\n')[-1])
        print(w, d['ms_per_step'], d['block_stats_ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    except Exception as e: print(w, 'ERR', e)
PY
