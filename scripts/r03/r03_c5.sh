#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; rm -rf $O; mkdir -p $O
GP_COV_FAST=0 timeout 300 python scripts/r03_c5.py --ref /tmp/cov_ref.npy 2>&1 | grep "^{" | tee -a $O/c5.jsonl
GP_KNN_DEBUG=1 GP_COV_FAST=1 GP_COV_FAST_WAVES=4 timeout 300 python scripts/r03_c5.py --ref /tmp/cov_ref.npy 2>&1 | grep "^{\|left" | tail -3 | tee -a $O/c5.jsonl
GP_COV_FAST=1 GP_COV_FAST_WAVES=4 timeout 300 python scripts/r03_c5.py --ref /tmp/cov_ref.npy 2>&1 | grep "^{" | tee -a $O/c5.jsonl
GP_COV_FAST=1 GP_COV_FAST_WAVES=3 timeout 300 python scripts/r03_c5.py --ref /tmp/cov_ref.npy 2>&1 | grep "^{" | tee -a $O/c5.jsonl
for mode in 0 1; do
rm -rf /tmp/pk && GP_COV_FAST=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python scripts/r03_c5.py > /tmp/c5_prof.log 2>&1
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats_fast$mode.csv && head -8 $f | cut -c1-200
done
timeout 900 python -m pytest tests/test_knn_gicp_gpu.py tests/test_configs_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest exit $?" >> $O/pytest.txt; tail -4 $O/pytest.txt | cut -c1-300
