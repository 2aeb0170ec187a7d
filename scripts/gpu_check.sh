#!/bin/bash
# Runs on the GPU box through gpurun: tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
tail -4 gpurun_out/bench.log
rm -rf gpurun_out/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-iters 20 > gpurun_out/rocprof.log 2>&1
find gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$ctr
  GP_BENCH_CALIBRATE=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-iters 5 > gpurun_out/pmc_$ctr.log 2>&1
done
ff=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_summary.py "$ff" "$fw" 1000000 gpurun_out/hbm_traffic.json | tee gpurun_out/pmc_summary.txt
