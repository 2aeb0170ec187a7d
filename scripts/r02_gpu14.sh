#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest -m gpu -x -q tests/test_knn_gicp_gpu.py tests/test_configs_gpu.py 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
GP_KNN_DEBUG=1 timeout 300 python scripts/r02_profile_aux.py c5 6 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8
