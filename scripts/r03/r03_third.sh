#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; rm -rf $O; mkdir -p $O
timeout 900 python scripts/r03_sweep.py 11:0:0:0,12:0:100:0,12:0:100:1,12:0:100:2,12:0:100:3,12:0:200:1,12:0:300:1,12:0:50:1 --big > $O/sweep.jsonl 2> $O/sweep.err; echo "sweep exit $?" >> $O/sweep.err; grep -v '"trace"' $O/sweep.jsonl | cut -c1-250
tail -3 $O/sweep.err
timeout 600 python -m pytest tests/test_vgicp_gpu.py tests/test_configs_gpu.py -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt | cut -c1-300
