#!/bin/bash
# functional rehearsal of the multi-rank bench path on a 1-GPU box: N ranks share the GPU, gloo instead of RCCL
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GP_BENCH_BACKEND=gloo
mkdir -p gpurun_out
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 50 --warmup 5 --no-cpu-baseline --c4-steps 5 > gpurun_out/r02_rehearsal_n$n.log 2>&1
  echo "rehearsal $n exit $?" >> gpurun_out/r02_rehearsal_n$n.log
  grep "^{" gpurun_out/r02_rehearsal_n$n.log | cut -c1-200; tail -1 gpurun_out/r02_rehearsal_n$n.log
done
