#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_vgicp_gpu.py -m gpu -q -x -k "fused or every_kernel or stream_kernel" > $O/pytest_a.txt 2>&1; echo "pytest exit $?" >> $O/pytest_a.txt; tail -5 $O/pytest_a.txt | cut -c1-300
timeout 600 python bench.py --no-configs --no-c4 > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log; grep "^{" $O/bench.log | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step']); print(json.dumps(r['roofline'], indent=0)[:3000])"; tail -2 $O/bench.log | cut -c1-300
