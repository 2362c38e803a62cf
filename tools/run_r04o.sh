TAG=${1:-r04o}; mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_lidar.py tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.log
for V in main nobatch; do
  L=""; [ $V != main ] && L="$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$V.so"
  NVBX_LIB=$L timeout 400 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/$TAG/bench_lidar_$V.json 2> gpurun_out/$TAG/bench_lidar_$V.err; echo "lidar $V rc=$?"
  NVBX_LIB=$L timeout 400 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_$V.json 2> gpurun_out/$TAG/bench_$V.err; echo "camera $V rc=$?"
done
python - <<PY
import json
for w in ('lidar_main','lidar_nobatch','main','nobatch'):
    try:
        d=json.loads(open('gpurun_out/$TAG/bench_%s.json' % w).read().strip().split('\n')[-1])
        print(w, d['ms_per_step'], d.get('ms_per_step_exploring'), d.get('ms_per_step_revisit'), d.get('ms_per_step_classic_order'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
        if 'timing' in d and 'exploring' in d['timing']: print('   ', str(d['timing']['exploring'])[:300])
    except Exception as e: print(w, 'ERR', e)
PY
