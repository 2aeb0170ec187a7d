#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_vgicp_gpu.py -q -m gpu -x > gpurun_out/r02_pytest2.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest2.log
tail -5 gpurun_out/r02_pytest2.log
timeout 900 python scripts/r02_sweep.py 4,5,6,7 0 --big > gpurun_out/r02_sweep2.jsonl 2> gpurun_out/r02_sweep2.err; echo "sweep exit $?"
tail -3 gpurun_out/r02_sweep2.err
for v in 4 5 6 7; do GP_VARIANT=$v timeout 600 python scripts/bench_configs.py C3,C4 2>gpurun_out/r02_cfg_v$v.err | cut -c1-420 | tee -a gpurun_out/r02_configs_variants.txt; done
