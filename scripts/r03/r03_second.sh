#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; rm -rf $O; mkdir -p $O
timeout 900 python scripts/r03_sweep.py 11:0:0,12:0:0,12:0:50,12:0:100,12:0:150,12:0:200,12:0:300,12:0:400 --big > $O/sweep.jsonl 2> $O/sweep.err; echo "sweep exit $?" >> $O/sweep.err; grep -v '"trace"' $O/sweep.jsonl | cut -c1-330
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -30 $O/pytest_gpu.txt | cut -c1-300
