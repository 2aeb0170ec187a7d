#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_vgicp_gpu.py -m gpu -q -x -k "fused or every_kernel or stream_kernel or non_finite" > $O/pytest_a.txt 2>&1; echo "pytest exit $?" >> $O/pytest_a.txt; tail -5 $O/pytest_a.txt | cut -c1-300
timeout 300 python scripts/r03_fused.py 1500 4 > $O/fused.jsonl 2> $O/fused.err; echo "fused exit $?" >> $O/fused.err; cut -c1-400 $O/fused.jsonl; tail -3 $O/fused.err
