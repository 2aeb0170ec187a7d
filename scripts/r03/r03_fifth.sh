#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_vgicp_gpu.py -m gpu -q -x -k "fused or every_kernel or stream_kernel" > $O/pytest_a.txt 2>&1; echo "pytest exit $?" >> $O/pytest_a.txt; tail -5 $O/pytest_a.txt | cut -c1-300
timeout 300 python scripts/r03_fused.py 1500 4 > $O/fused.jsonl 2> $O/fused.err; echo "fused exit $?" >> $O/fused.err; cat $O/fused.jsonl; tail -3 $O/fused.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o fused -- python $GRAFT_REPO_ROOT/scripts/r03_fused.py 300 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -8 {} | cut -c1-260'
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
