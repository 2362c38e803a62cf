# A/B of LiDAR variants inside one GPU session: tools/lidar_variant_ab.sh TAG variant [variant ...]   (variants: tools/build_variant.sh NAME "-D...")
TAG=$1; shift; mkdir -p gpurun_out/$TAG
for X in main "$@"; do
  LL=""; [ $X != main ] && LL=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$X.so
  if [ $X != main ]; then NVBX_LIB=$LL timeout 600 python -m pytest tests/test_lidar.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/$TAG/pytest_$X.log 2>&1; echo "pytest $X rc=$?"; fi
  NVBX_LIB=$LL timeout 400 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/$TAG/bench_lidar_$X.json 2> gpurun_out/$TAG/bench_lidar_$X.err; echo "lidar $X rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/$TAG/bench_lidar_$X.json').read().strip().split('\n')[-1])
print('$X', d['ms_per_step'], d.get('ms_per_step_exploring'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
PY
done
