TAG=$1; V=$2; mkdir -p gpurun_out/$TAG
L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$V.so
NVBX_LIB=$L timeout 600 python -m pytest tests/test_lidar.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/$TAG/pytest.log
for X in main $V; do
  LL=""; [ $X != main ] && LL=$L
  NVBX_LIB=$LL timeout 400 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/$TAG/bench_lidar_$X.json 2> gpurun_out/$TAG/bench_lidar_$X.err; echo "lidar $X rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/$TAG/bench_lidar_$X.json').read().strip().split('\n')[-1])
print('$X', d['ms_per_step'], d.get('ms_per_step_exploring'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
PY
done
