#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_configs_gpu.py tests/test_vgicp_gpu.py tests/test_knn_gicp_gpu.py -q -m gpu -x --durations=8 > gpurun_out/r02_pytest3.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest3.log
tail -25 gpurun_out/r02_pytest3.log
timeout 900 python scripts/r02_sweep.py 4,5,6,7 0 > gpurun_out/r02_sweep3.jsonl 2> gpurun_out/r02_sweep3.err; echo "sweep exit $?"
tail -3 gpurun_out/r02_sweep3.err
for v in 4; do GP_VARIANT=$v timeout 600 python scripts/bench_configs.py C1,C3,C4 2>gpurun_out/r02_cfg3_v$v.err | cut -c1-420 | tee -a gpurun_out/r02_configs_variants3.txt; done
