# round-6 A/B inside one GPU session: tools/ab_r6.sh TAG "bench args" v0 v1 ...  (camera workload; prints ms_per_step + kernel averages; then timelines of *wgt variants)
TAG=$1; ARGS=$2; shift 2
mkdir -p gpurun_out/$TAG
for rep in 1 2; do for v in "$@"; do
  case $v in *wgt) continue;; esac
  L=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so; [ $v = current ] && L=$PWD/isaac_ros_nvblox_amd/libnvblox_hip.so
  NVBX_LIB=$L timeout 300 python bench.py $ARGS --no-cpu-baseline > gpurun_out/$TAG/cam_${v}_$rep.json 2>gpurun_out/$TAG/cam_${v}_$rep.err
  python - <<PY
import json
j=json.load(open("gpurun_out/$TAG/cam_${v}_$rep.json"))
print("$v:", j["ms_per_step"], j.get("ms_per_step_revisit"), j["parity"]["ok"], j["frame_latency"]["wall_ms"]["p50"], {k: round(x["avg_us"],2) for k,x in j["kernels"].items()})
PY
done; done
for v in "$@"; do
  case $v in *wgt) ;; *) continue;; esac
  NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$v.so python tools/wg_timeline.py 2>/dev/null > gpurun_out/$TAG/$v.json
  python - <<PY
import json
d=json.load(open('gpurun_out/$TAG/$v.json'))
mv=d['k_mark_view']
print("== $v")
for k in ('tiles','trace','mark'):
    r=dict(mv[k]); sl=r.pop('slowest'); r.pop('rounds_hist',None); print(k, json.dumps(r)[:900])
    for s in sl[:2]: print('    ', s)
print('rounds', mv['trace'].get('rounds_hist'))
print('launch ends', mv['launch_end'], d['k_integrate_tsdf_color']['launch_end'])
PY
done
