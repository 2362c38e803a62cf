#!/bin/bash
# One GPU-box round: parity tests, bench, rocprofv3 kernel stats (+ optional PMC passes).  Usage: tools/gpu_round.sh TAG [pmc]
TAG=${1:-rXX}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/$TAG/pytest.log
tail -5 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
cat gpurun_out/$TAG/bench.json
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -o stats -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/$TAG/prof_bench.json 2> $R/gpurun_out/$TAG/prof.err); echo "rocprof rc=$?"
if [ "$2" = "pmc" ]; then
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -o pmc -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/$TAG/pmc_fetch.err); echo "pmc fetch rc=$?"
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -o pmc -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/$TAG/pmc_write.err); echo "pmc write rc=$?"
fi
find gpurun_out/$TAG -name "*.csv" | head -20
find gpurun_out/$TAG -name "*kernel_stats.csv" -exec cat {} \;
# LiDAR workload (configs[4]): bench line + kernel stats
timeout 600 python bench.py --workload lidar --steps 100 --warmup 10 > gpurun_out/$TAG/bench_lidar.json 2> gpurun_out/$TAG/bench_lidar.err; echo "lidar bench rc=$?"; cat gpurun_out/$TAG/bench_lidar.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats_lidar -o stats -- python $R/bench.py --workload lidar --steps 50 --warmup 5 > /dev/null 2> $R/gpurun_out/$TAG/prof_lidar.err); echo "rocprof lidar rc=$?"
find gpurun_out/$TAG/stats_lidar -name "*kernel_stats.csv" -exec cat {} \;
