#!/bin/bash
# One GPU-box round: (parity tests,) bench line + rocprofv3 kernel stats (+ PMC FETCH/WRITE passes) for every workload.
# Usage: tools/gpu_round.sh TAG [pmc] [notest] [workloads...]   (default workloads: camera lidar decay multicam)
TAG=${1:-rXX}; shift
PMC=0; TEST=1; WL=()
for a in "$@"; do case $a in pmc) PMC=1;; notest) TEST=0;; *) WL+=($a);; esac; done
[ ${#WL[@]} -eq 0 ] && WL=(camera lidar decay multicam)
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
R=$PWD
if [ $TEST = 1 ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$TAG/pytest.log
  tail -5 gpurun_out/$TAG/pytest.log
fi
for W in "${WL[@]}"; do
  S=""; [ $W != camera ] && S="_$W"
  case $W in lidar) ARGS="--steps 100 --warmup 10"; PARGS="--steps 50 --warmup 5 --profile-run";; decay) ARGS="--steps 120 --warmup 24"; PARGS="--steps 60 --warmup 12";;
             node) ARGS=""; PARGS="--no-parity";;
             camera_mesh) ARGS="--steps 100 --warmup 20 --with-mesh"; PARGS="--steps 100 --warmup 20 --profile-run --with-mesh";;
             camera_zc) ARGS="--zero-copy-deferral"; PARGS="--steps 100 --warmup 20 --profile-run --zero-copy-deferral";;
             camera_k20) ARGS="--steps 20 --warmup 5"; PARGS="--steps 20 --warmup 5 --profile-run";;
             multicam) ARGS="--steps 100 --warmup 20 --cameras 4"; PARGS="--steps 50 --warmup 10 --cameras 4 --profile-run";; multicam8) ARGS="--steps 100 --warmup 20 --cameras 8"; PARGS="--steps 50 --warmup 10 --cameras 8 --profile-run";; *) ARGS=""; PARGS="--steps 100 --warmup 20 --profile-run";; esac
  WL_NAME=$W; [ $W = multicam8 ] && WL_NAME=multicam; [ $W = camera_mesh ] && WL_NAME=camera; [ $W = camera_zc ] && WL_NAME=camera; [ $W = camera_k20 ] && WL_NAME=camera
  PMS=100; [ $W = lidar ] && PMS=500     # (LiDAR: few, long launches -- a longer run keeps the first-launch outliers out of the average)
  timeout 900 python bench.py --workload $WL_NAME $ARGS > gpurun_out/$TAG/bench$S.json 2> gpurun_out/$TAG/bench$S.err; echo "bench $W rc=$?"
  cat gpurun_out/$TAG/bench$S.json | cut -c1-600
  # (profiling runs: 100 ms timed instead of 1 s -- the per-launch traces / counter tables of a 1 s run are tens of MB, gpurun merges <= 64 MiB)
  (cd /tmp && NVBX_BENCH_MIN_MS=$PMS timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats$S -o stats -- python $R/bench.py --workload $WL_NAME $PARGS --no-cpu-baseline > $R/gpurun_out/$TAG/prof_bench$S.json 2> $R/gpurun_out/$TAG/prof$S.err); echo "rocprof $W rc=$?"
  find gpurun_out/$TAG/stats$S -name "*kernel_trace.csv" -delete
  find gpurun_out/$TAG/stats$S -name "*kernel_stats.csv" -exec head -12 {} \;
  if [ $PMC = 1 ]; then
    (cd /tmp && NVBX_BENCH_MIN_MS=30 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch$S -o pmc -- python $R/bench.py --workload $WL_NAME $PARGS --no-cpu-baseline > /dev/null 2> $R/gpurun_out/$TAG/pmc_fetch$S.err); echo "pmc fetch $W rc=$?"
    (cd /tmp && NVBX_BENCH_MIN_MS=30 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write$S -o pmc -- python $R/bench.py --workload $WL_NAME $PARGS --no-cpu-baseline > /dev/null 2> $R/gpurun_out/$TAG/pmc_write$S.err); echo "pmc write $W rc=$?"
  fi
done
