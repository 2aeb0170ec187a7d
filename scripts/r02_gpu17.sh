#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r02_pytest_gpu.log
timeout 600 python bench.py --no-c4 2>&1 | grep "^{" | tee gpurun_out/r02_bench5.log
timeout 300 python scripts/r02_profile_aux.py c5 10 2>&1 | grep "C5 1M"
