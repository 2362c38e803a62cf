TAG=${1:-r04q}; mkdir -p gpurun_out/$TAG
NVBX_COLOR_DEFERRAL=2 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/$TAG/pytest_default_staged.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/$TAG/pytest_default_staged.log
