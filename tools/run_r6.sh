# round-6 GPU call: (optionally) the GPU suite, then the driver-flag camera line.   usage: tools/run_r6.sh TAG [full|quick|none] [extra bench workloads...]
TAG=${1:-r6}; MODE=${2:-quick}; shift; shift
mkdir -p gpurun_out/$TAG
if [ "$MODE" = full ]; then T="tests"; elif [ "$MODE" = quick ]; then T="tests/test_gpu_pipeline.py tests/test_gpu_sequences.py tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_frames.py tests/test_cpp_facade.py"; else T=""; fi
if [ -n "$T" ]; then timeout 1500 python -m pytest $T -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/$TAG/pytest.log; fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_k20.json 2> gpurun_out/$TAG/bench_k20.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/$TAG/bench_k20.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_revisit','ms_per_step_classic_order')}, d['parity']['ok'], d['frame_latency']['wall_ms'])
print({k:(v['avg_us'],v['launches_per_step']) for k,v in d['kernels'].items()})
PY
for W in "$@"; do
  case $W in lidar) ARGS="--steps 100 --warmup 10";; decay) ARGS="--steps 120 --warmup 24";; multicam) ARGS="--steps 100 --warmup 20 --cameras 4";; multicam8) ARGS="--steps 100 --warmup 20 --cameras 8";; *) ARGS="";; esac
  WN=$W; [ $W = multicam8 ] && WN=multicam
  timeout 900 python bench.py --workload $WN $ARGS --no-cpu-baseline > gpurun_out/$TAG/bench_$W.json 2> gpurun_out/$TAG/bench_$W.err; echo "bench $W rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/$TAG/bench_$W.json')); print('$W', d['value'], d['unit'], d['ms_per_step'], (d.get('parity') or {}).get('ok')); print({k:(v['avg_us'],v['launches_per_step']) for k,v in d.get('kernels',{}).items()})"
done
