#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_vgicp_gpu.py -m gpu -q -x -k "overlapped or every_kernel or ragged or second_and_third" > $O/pytest_a.txt 2>&1; echo "pytest exit $?" >> $O/pytest_a.txt; tail -5 $O/pytest_a.txt | cut -c1-300
timeout 600 python scripts/r03_sweep.py 12:0:100:1:0,12:0:100:1:1,12:0:200:1:1,12:0:200:1:0 --no-trace > $O/sweep.jsonl 2> $O/sweep.err; echo "sweep exit $?" >> $O/sweep.err; grep -v '"trace"' $O/sweep.jsonl | cut -c1-330
tail -3 $O/sweep.err
timeout 600 python bench.py --no-configs > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log; grep "^{" $O/bench.log | cut -c1-2200; tail -3 $O/bench.log | cut -c1-300
