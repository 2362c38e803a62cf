TAG=${1:-r04t}; mkdir -p gpurun_out/$TAG
V=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_laneord.so
NVBX_LIB=$V timeout 900 python -m pytest tests/test_lidar.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest.log
for X in main laneord; do
  L=""; [ $X != main ] && L=$V
  NVBX_LIB=$L timeout 400 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/$TAG/bench_lidar_$X.json 2> gpurun_out/$TAG/bench_lidar_$X.err; echo "lidar $X rc=$?"
done
python - <<PY
import json
for w in ('main','laneord'):
    try:
        d=json.loads(open('gpurun_out/$TAG/bench_lidar_%s.json' % w).read().strip().split('\n')[-1])
        print(w, d['ms_per_step'], d.get('ms_per_step_exploring'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    except Exception as e: print(w, 'ERR', e)
PY
