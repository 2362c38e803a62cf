#!/bin/bash
# k_mesh register budget (waves per SIMD it is compiled for: __launch_bounds__(512, NVBX_MESH_WAVES)).  LDS (41 KiB per workgroup) allows 3
# workgroups = 6 waves per SIMD; 124 VGPRs allow 4.  Build step (here, no GPU): tools/mesh_occupancy_sweep.sh build  -> variants/libnvblox_hip_meshN.so
# Measure (GPU box, one session): tools/mesh_occupancy_sweep.sh run
cd "$(dirname "$0")/.."
CS=isaac_ros_nvblox_amd/csrc
if [ "$1" = build ]; then
  mkdir -p isaac_ros_nvblox_amd/variants
  make -C $CS -j8 > /dev/null
  for w in 2 4 5 6; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DNVBX_MESH_WAVES=$w -c $CS/mesh.hip -o /tmp/mesh_$w.o
    OBJS=$(ls $CS/*.o | grep -v "/mesh.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o isaac_ros_nvblox_amd/variants/libnvblox_hip_mesh$w.so $OBJS /tmp/mesh_$w.o
  done
  ls -la isaac_ros_nvblox_amd/variants
  exit 0
fi
for w in 2 4 5 6; do
  echo "NVBX_MESH_WAVES=$w"
  NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_mesh$w.so timeout 300 python tools/maintenance_bw.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   full-layer mesh on the LiDAR map:', {k: v['avg_us'] for k, v in d['kernels'].items() if 'mesh' in k})"
  NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_mesh$w.so NVBX_BENCH_MIN_MS=200 timeout 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('   per-frame k_mesh us:', d['kernels']['k_mesh']['avg_us'], ' mesh component ms:', d['ms_components'].get('mesh'))"
done
