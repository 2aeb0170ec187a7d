#!/bin/bash
# Round 5 (VERDICT r04 #5b): the N > 1 launch of bench.py rehearsed on ONE GPU -- torch.distributed with the gloo backend (RCCL needs one device per rank), every rank on cuda:0.
# Exactly the driver's command for N = 2, 4, 8; checks that the job prints ONE JSON line (rank 0's) and that the line names its exchanges.  Logs: gpurun_out/r05_rehearsal/n<N>.log
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GP_BENCH_BACKEND=gloo
O=gpurun_out/r05_rehearsal; mkdir -p $O
for N in ${1:-2 4 8}; do
  P=$((29600 + N))
  ( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $P bench.py --gpus $N --steps 20 --warmup 5 ) > $O/n$N.log 2> $O/n$N.err
  echo "exit $?" >> $O/n$N.log
  python - $O/n$N.log $N <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
n = int(sys.argv[2])
assert len(lines) == 1, f"expected ONE JSON line, found {len(lines)}"
d = json.loads(lines[0])
assert d["n_gpus"] == n and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
c4 = d["c4"]
print(json.dumps(dict(n=n, value=d["value"], ms_per_step=d["ms_per_step"], ms_per_step_cold=d["ms_per_step_cold"], exchange=d["config"]["exchange"], c4_exchange=c4["exchange"],
                      c4_ms=c4["ms_per_linearize"], c4_allreduce_ms=c4["allreduce_ms"], c4_allgather_ms=c4["allgather_ms"],
                      c4_inlib={k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("exchange", "ms", "records_equal_first_leg") if kk in v}) for k, v in (c4.get("inlib") or {}).items()
                                if k in ("devices", "error", "all_gather", "all_reduce", "no_collective")})))
PY
done
