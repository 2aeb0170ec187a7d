#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r02_pytest_gpu.log
GP_KNN_DEBUG=1 timeout 300 python scripts/r02_profile_aux.py c5 6 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
timeout 300 python scripts/r02_profile_aux.py map 10 2>&1 | grep "voxel map"
