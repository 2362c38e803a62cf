#!/bin/bash
# workgroups of the TSDF part (NVBX_INTEG_GRID) and the colour part (NVBX_COLOR_GRID) of the fused launch, camera batches: tools/fused_grid_sweep.sh CAMERAS "T:C T:C ..."
C=${1:-8}; shift
for P in ${@:-1024:1024 768:768 512:512 384:384 640:384 384:640 512:1024 1024:512}; do
  T=${P%%:*}; K=${P#*:}
  NVBX_INTEG_GRID=$T NVBX_COLOR_GRID=$K timeout 300 python bench.py --workload multicam --cameras $C --steps 100 --warmup 20 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cams $C tsdf $T colour $K:', d['ms_per_step'], {k: round(x['avg_us'],1) for k,x in d['kernels'].items() if k in ('k_mark_view','k_integrate_tsdf_color')})"
done
