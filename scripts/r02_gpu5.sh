#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_multi_gpu.py tests/test_host_gpu.py tests/test_voxelmap_gpu.py tests/test_cloud_gpu.py tests/test_solver_gpu.py -q -m gpu --durations=6 > gpurun_out/r02_pytest5.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest5.log
tail -30 gpurun_out/r02_pytest5.log
