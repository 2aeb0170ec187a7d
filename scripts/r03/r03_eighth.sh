#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_vgicp_gpu.py -m gpu -q -x > $O/pytest_a.txt 2>&1; echo "pytest exit $?" >> $O/pytest_a.txt; tail -5 $O/pytest_a.txt | cut -c1-300
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_multi_gpu.py tests/test_host_gpu.py -m gpu -q -x > $O/pytest_b.txt 2>&1; echo "pytest exit $?" >> $O/pytest_b.txt; tail -5 $O/pytest_b.txt | cut -c1-300
for wl in c4 c3; do
  C4_CONFIGS=0:0:0 WORKLOAD=$wl timeout 300 python scripts/r03_c4_traffic.py 2>/dev/null | grep "^{" | head -1 | cut -c1-300 | tee -a $O/times.jsonl
done
