cd $GRAFT_REPO_ROOT; make -C tests/cpp round6_checks >/dev/null 2>&1
python - <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import helpers as H
from test_cpp_facade import _write_frames_bin
cam = H.SMALL_CAM
_write_frames_bin("/tmp/frames.bin", H.frames(6, cam, color=True, stride=9), cam)
PY
bad=0
for i in $(seq 1 60); do
  out=$(LD_LIBRARY_PATH=$PWD/isaac_ros_nvblox_amd:$LD_LIBRARY_PATH tests/cpp/round6_checks lifetime /tmp/frames.bin 2>&1 | tail -1); rc=$?
  case "$out" in *'"busy_frames": 6, "handed_out_while_busy": 0, "equal": true, "reused_when_idle": true'*) ;; *) bad=$((bad+1)); echo "run $i: $out";; esac
done
echo "not all-good runs: $bad of 60"
