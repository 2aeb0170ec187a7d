#!/bin/bash
# k-NN / GICP tests + C5 timing of the row-tiled covariance kernel vs the per-lane search
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -x -q tests/test_knn_gicp_gpu.py tests/test_configs_gpu.py 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r02_pytest_knn.log
GP_KNN_DEBUG=1 timeout 300 python scripts/r02_profile_aux.py counters 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r02_c5_counters.jsonl
cd /tmp
for mode in 0 2; do
  rm -rf /tmp/pk
  GP_KNN_MODE=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py c5 10 > /tmp/pk.log 2>&1
  grep "C5 1M" /tmp/pk.log | sed "s/^/mode $mode: /" | tee -a $GRAFT_REPO_ROOT/gpurun_out/r02_c5_wall.txt
  f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r02_c5_kernel_stats_mode$mode.csv
  head -8 $f | cut -c1-150
done
