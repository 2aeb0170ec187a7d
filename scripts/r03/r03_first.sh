#!/bin/bash
# Round-3 first GPU pass: whole GPU test-suite, kernel-family sweep with the per-workgroup timeline, default bench.  Outputs under gpurun_out/r03a/.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; rm -rf $O; mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt; nproc >> $O/device.txt
timeout 600 python scripts/r03_sweep.py 8:0:1,11:0:1,12:0:1,12:0:0,12:1:1 --big > $O/sweep.jsonl 2> $O/sweep.err; echo "sweep exit $?" >> $O/sweep.err; cut -c1-420 $O/sweep.jsonl | grep -v '"trace"'
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log; grep "^{" $O/bench.log > $O/bench_n1.json; cut -c1-1500 $O/bench_n1.json; tail -5 $O/bench.log | cut -c1-600
