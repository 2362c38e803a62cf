# quick GPU round: the pipeline / sequence / ESDF tests, the per-workgroup timeline, the driver-flag bench line.   usage: tools/run_quick.sh TAG [full]
TAG=${1:-rq}; mkdir -p gpurun_out/$TAG
if [ "$2" = full ]; then T="tests"; else T="tests/test_gpu_pipeline.py tests/test_gpu_sequences.py tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_batch.py"; fi
timeout 1500 python -m pytest $T -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_wgt.so python tools/wg_timeline.py 2>/dev/null > gpurun_out/$TAG/wgt.json; echo "wgt rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_k20.json 2> gpurun_out/$TAG/bench_k20.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/$TAG/bench_k20.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_revisit','ms_per_step_classic_order')}, d['parity']['ok'])
print({k:(v['avg_us'],v['launches_per_step']) for k,v in d['kernels'].items()})
d=json.load(open('gpurun_out/$TAG/wgt.json'))
mv=d['k_mark_view']
for k in ('tiles','trace','scan','mark'):
    r=dict(mv[k]); sl=r.pop('slowest'); r.pop('rounds_hist',None); print(k, json.dumps(r))
    for s in sl[:3]: print('    ', s)
print('launch ends', mv['launch_end'], d['k_integrate_tsdf_color']['launch_end'], {k:(v['dur_median'],v['dur_max']) for k,v in d['k_integrate_tsdf_color'].items() if isinstance(v,dict)})
PY
