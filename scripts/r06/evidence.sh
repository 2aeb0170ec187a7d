#!/bin/bash
# Round 6 evidence: the driver's bench command three times (outside clock), rocprofv3 --kernel-trace --stats of the same command, the N = 2 rehearsal (gloo on one GPU).
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r06; mkdir -p $O; R=$PWD
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt; nproc >> $O/device.txt
RUNS="${RUNS:-1 2 3}" NO_REHEARSAL=1 bash scripts/r06/bench_check.sh
rm -rf /tmp/bprof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bprof -o b -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --detail-file /tmp/bench_detail_prof.json > $O/bench_under_rocprof.out 2> $O/bench_under_rocprof.err)
f=$(find /tmp/bprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -4 $f | cut -c1-220
for N in 2; do
  ( time GP_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 20 --warmup 5 ) > $O/rehearsal_n$N.out 2> $O/rehearsal_n$N.err
  echo "rehearsal N=$N exit $?"; tail -n 1 $O/rehearsal_n$N.out | cut -c1-1200; cp bench_detail.json $O/bench_detail_rehearsal_n$N.json 2>/dev/null
done
