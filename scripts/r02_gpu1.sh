#!/bin/bash
# round 2, GPU call 1: the GPU test-suite on the new lookup structure + the tile-kernel sweep
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -q -m gpu -x --deselect tests/test_solver_gpu.py > gpurun_out/r02_pytest1.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest1.log
tail -15 gpurun_out/r02_pytest1.log
timeout 900 python scripts/r02_sweep.py 1,2,3,4 0,1,2,3,4,6 --big > gpurun_out/r02_sweep.jsonl 2> gpurun_out/r02_sweep.err; echo "sweep exit $?"
cat gpurun_out/r02_sweep.jsonl | cut -c1-400
tail -5 gpurun_out/r02_sweep.err
