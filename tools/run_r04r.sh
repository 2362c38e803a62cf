TAG=${1:-r04r}; mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.log
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "camera rc=$?"
timeout 400 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/$TAG/bench_k20.json 2> gpurun_out/$TAG/bench_k20.err; echo "camera k20 rc=$?"
timeout 400 python bench.py --no-cpu-baseline --zero-copy-deferral > gpurun_out/$TAG/bench_zc.json 2> gpurun_out/$TAG/bench_zc.err; echo "camera zc rc=$?"
timeout 400 python bench.py --workload decay --steps 120 --warmup 24 --no-cpu-baseline > gpurun_out/$TAG/bench_decay.json 2> gpurun_out/$TAG/bench_decay.err; echo "decay rc=$?"
python - <<PY
import json
for w in ('','_k20','_zc','_decay'):
    try:
        d=json.loads(open('gpurun_out/$TAG/bench%s.json' % w).read().strip().split('\n')[-1])
        print(w or 'camera', d['ms_per_step'], d.get('ms_per_step_revisit'), d.get('ms_per_step_classic_order'), d.get('color_deferral',{}).get('form'), {k:v for k,v in d.get('color_deferral',{}).items() if k.startswith('ms_')}, (d.get('parity') or {}).get('ok'))
        print('   ', {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    except Exception as e: print(w, 'ERR', e)
PY
