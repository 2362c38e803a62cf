#!/bin/bash
# PMC pass for latency analysis: wave lifetime and wait split per kernel.  Usage: tools/gpu_pmc.sh TAG "COUNTERS..."
TAG=${1:-pmcX}; shift
R=$PWD; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
(cd /tmp && timeout ${PMC_TIMEOUT:-600} rocprofv3 --pmc $@ --output-format csv -d $R/gpurun_out/$TAG -o pmc -- python $R/bench.py ${BENCH_ARGS:---steps 30 --warmup 10 --no-cpu-baseline} > /dev/null 2> $R/gpurun_out/$TAG/err.txt); echo "rc=$?"
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/$TAG/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:20]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in sorted(acc.items()):
    print(k, {c: round(v[0] / v[1], 1) for c, v in d.items()})
PY
