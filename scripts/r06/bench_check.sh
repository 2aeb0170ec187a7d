#!/bin/bash
# Round 6: the driver's bench command as the driver runs it (timed from outside), the last line checked the way the driver parses it, the detail file kept;
# then the N = 2 rehearsal on one GPU (gloo: RCCL needs a device per rank) and the one-rank RCCL test.  Outputs: gpurun_out/r06/
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06; mkdir -p $O
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt; nproc >> $O/device.txt
for i in ${RUNS:-1 2}; do
  S=$(date +%s.%N)
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_run$i.out 2> $O/bench_run$i.err; rc=$?
  E=$(date +%s.%N)
  tail -n 1 $O/bench_run$i.out > $O/bench_run$i.json
  cp bench_detail.json $O/bench_detail_run$i.json 2>/dev/null
  python3 - $O/bench_run$i.json $rc $S $E <<'PY'
import json, sys
line = open(sys.argv[1]).read().rstrip("\n")
d = json.loads(line)
print(json.dumps(dict(rc=int(sys.argv[2]), wall_s=round(float(sys.argv[4]) - float(sys.argv[3]), 1), bytes=len(line.encode()), ms_per_step=d["ms_per_step"], cold=d["ms_per_step_cold"],
                      frac=d["roofline"]["frac"], frac_rocprof=d["roofline"]["frac_rocprof"], rocprof_avg_ms=d["roofline"]["rocprof_avg_ms"], traffic=d["roofline"]["traffic"],
                      cpu=d["cpu_baseline"] and d["cpu_baseline"]["ms_per_linearize"], parity=d["parity_max"], legs=d["legs"], skipped=d["legs_skipped"], run_seconds=d["run_seconds"])))
PY
done
if [ -z "${NO_REHEARSAL:-}" ]; then
  for N in ${REHEARSE:-2}; do
    ( time GP_BENCH_BACKEND=gloo timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 20 --warmup 5 ) > $O/rehearsal_n$N.out 2> $O/rehearsal_n$N.err
    echo "exit $?"; tail -n 1 $O/rehearsal_n$N.out | cut -c1-1500; cp bench_detail.json $O/bench_detail_rehearsal_n$N.json 2>/dev/null
  done
  timeout 900 python3 -m pytest tests/test_multi_gpu.py tests/test_peer_exchange_gpu.py -q -m gpu -x 2>&1 | tail -5
fi
