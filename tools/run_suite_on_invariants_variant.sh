#!/bin/bash
# The GPU parity / pipeline / sequence tests once more on the -DNVBX_CHECK_INVARIANTS variant of the library: every mapper reports its violation
# counters when it is closed (NVBX_CHECK_ON_CLOSE=1; isaac_ros_nvblox_amd/mapper.py INVARIANT_REPORT), the report is printed at the end of the session.
cd "$(dirname "$0")/.."
[ -f isaac_ros_nvblox_amd/variants/libnvblox_hip_inv.so ] || bash tools/build_variant.sh inv "-DNVBX_CHECK_INVARIANTS"
cat > /tmp/inv_report_plugin.py <<'PY'
import gc
def pytest_sessionfinish(session, exitstatus):
    gc.collect()
    from isaac_ros_nvblox_amd import mapper as M
    print("\nINVARIANT_REPORT", M.INVARIANT_REPORT)
PY
NVBX_LIB=$PWD/isaac_ros_nvblox_amd/variants/libnvblox_hip_inv.so NVBX_CHECK_ON_CLOSE=1 PYTHONPATH=/tmp:$PYTHONPATH \
  python -m pytest -p inv_report_plugin tests/test_gpu_pipeline.py tests/test_gpu_sequences.py tests/test_gpu_batch.py tests/test_gpu_frames.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_round6.py -m gpu -q "$@" 2>&1 | tail -8
