cd $GRAFT_REPO_ROOT
run() { L=""; [ "$1" != current ] && L="NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_$1.so"
  env $L NVBX_DECAY_GRID=$2 timeout 200 python tools/maintenance_bw.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1 grid $2:', {k:(v['avg_us'], v.get('achieved_GBps')) for k,v in j['kernels'].items() if 'decay' in k})"; }
for g in 768 1536 2048 3072; do run current $g; done
for g in 1024 2048 4096 8192; do run dw8 $g; done
