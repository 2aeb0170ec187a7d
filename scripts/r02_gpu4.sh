#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu -x --durations=6 > gpurun_out/r02_pytest4.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest4.log
tail -14 gpurun_out/r02_pytest4.log
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/r02_bench4.log 2>&1; echo "bench exit $?" >> gpurun_out/r02_bench4.log
tail -2 gpurun_out/r02_bench4.log | cut -c1-1800
for n in 2 4; do
  GP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02_rehearsal_n$n.log 2>&1; echo "rehearsal $n exit $?" >> gpurun_out/r02_rehearsal_n$n.log
  tail -2 gpurun_out/r02_rehearsal_n$n.log | cut -c1-1500
done
