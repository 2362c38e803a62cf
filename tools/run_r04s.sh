TAG=${1:-r04s}; mkdir -p gpurun_out/$TAG
for C in 4 8; do
  timeout 400 python bench.py --workload multicam --cameras $C --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/$TAG/bench_multicam$C.json 2> gpurun_out/$TAG/bench_multicam$C.err; echo "multicam $C rc=$?"
  timeout 400 python bench.py --workload multicam --cameras $C --steps 100 --warmup 20 --no-cpu-baseline --zero-copy-deferral > gpurun_out/$TAG/bench_multicam${C}_zc.json 2> gpurun_out/$TAG/bench_multicam${C}_zc.err; echo "multicam $C zc rc=$?"
done
python - <<PY
import json
for w in ('4','4_zc','8','8_zc'):
    try:
        d=json.loads(open('gpurun_out/$TAG/bench_multicam%s.json' % w).read().strip().split('\n')[-1])
        print(w, d['ms_per_step'], d.get('ms_per_step_revisit'), d.get('ms_per_step_classic_order'), d['value'], {k:v for k,v in d.get('color_deferral',{}).items() if k.startswith('ms_')}, (d.get('parity') or {}).get('ok'))
        print('   ', {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
        if 'batch_sweep' in d: print('   ', json.dumps(d['batch_sweep'])[:400])
    except Exception as e: print(w, 'ERR', e)
PY
