mkdir -p gpurun_out/r04g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04g/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04g/pytest.log
NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_wgt.so python tools/wg_timeline_lidar.py 2>/dev/null | head -1 | cut -c1-900
timeout 300 python bench.py --workload lidar --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null > gpurun_out/r04g/bench_lidar.json; python -c "import json; d=json.load(open('gpurun_out/r04g/bench_lidar.json')); print('lidar', d['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/r04g/bench_k20.json; python -c "import json; d=json.load(open('gpurun_out/r04g/bench_k20.json')); print('camera', d['ms_per_step'], d['parity']['ok'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
