#!/bin/bash
# XCD-affine record numbering (xcd_chunked, nvbx_internal.h): time and FETCH_SIZE of the variants in ONE box session.   usage: tools/chunk_ab.sh TAG v0 v1 ...
cd "$(dirname "$0")/.."
TAG=$1; shift; mkdir -p gpurun_out/$TAG
[ -z "$NO_TIMES" ] && for rep in 1 2; do bash tools/variant_ab_multicam.sh $TAG "$@"; done
for v in "$@"; do
  L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so; [ $v = current ] && L=$PWD/isaac_ros_nvblox_amd/libnvblox_hip.so
  echo "== FETCH_SIZE (KiB per launch; one counter per pass: two in one pass exceed the hardware and the profiler aborts), camera, $v"
  NVBX_LIB=$L NVBX_BENCH_MIN_MS=30 BENCH_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-parity --profile-run" bash tools/gpu_pmc.sh $TAG/pmc_cam_$v FETCH_SIZE | grep -v "^rc"
  echo "== the same, 8 cameras, $v"
  NVBX_LIB=$L NVBX_BENCH_MIN_MS=30 BENCH_ARGS="--workload multicam --cameras 8 --steps 50 --warmup 10 --no-cpu-baseline --no-parity --profile-run" bash tools/gpu_pmc.sh $TAG/pmc_m8_$v FETCH_SIZE | grep -v "^rc"
done
find gpurun_out/$TAG -name "*counter_collection.csv" -size +5M -delete
