#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -x -q tests/test_solver_gpu.py 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
timeout 600 python scripts/solver_time.py 2>&1 | grep "^{" | tee gpurun_out/r02_solver_time.jsonl | cut -c1-700
