#!/bin/bash
# full GPU suite + C5 work counters + C5 timings with the final default structure
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r02_pytest_gpu.log
timeout 300 python scripts/r02_profile_aux.py counters | tee gpurun_out/r02_c5_counters.jsonl
cd /tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py c5 10 > /tmp/pk.log 2>&1
grep "C5 1M" /tmp/pk.log | tee $GRAFT_REPO_ROOT/gpurun_out/r02_c5_wall.txt
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/r02_c5_kernel_stats.csv
head -12 $f | cut -c1-160
