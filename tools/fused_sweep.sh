#!/bin/bash
# A/B knobs of the two-launch camera pipeline inside ONE box session (boxes differ by a few %): lanes per ray of the sphere tracing that rides
# in the view-marking launch, riders before / after the tiles, fused launch on / off.
run() { echo "$*"; env "$@" NVBX_BENCH_MIN_MS=300 timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['ms_per_step'], d['ms_per_step_revisit'], {k:v['avg_us'] for k,v in d['kernels'].items()})"; }
run NVBX_FUSE_COLC=1
run NVBX_FUSED_TRACE_LANES=4
run NVBX_MARK_TILES_FIRST=0
run NVBX_FUSED_TRACE_LANES=4 NVBX_MARK_TILES_FIRST=0
run NVBX_FUSE_COLC=0
