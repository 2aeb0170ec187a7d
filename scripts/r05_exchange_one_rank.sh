# Round 5: what the exchange step costs around the 1 M-point step with ONE rank (nothing crosses a link: the fixed cost of each form) -- bench.py's N > 1 code path
# (GP_BENCH_FORCE_DIST=1, backend nccl = RCCL) with the peer exchange (one more kernel, the stack written to pinned host memory by it), the in-place ncclAllGather and the
# ncclAllReduce (both + a D2H copy), against the plain N = 1 step.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05x; mkdir -p $O
cd $GRAFT_REPO_ROOT
F="--steps 200 --warmup 20 --no-configs --no-c4 --no-traffic --no-big-source --cpu-seconds 1"
timeout 300 python bench.py $F > $O/plain.log 2>&1
for ex in peer all_gather all_reduce; do
  GP_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29700 + RANDOM % 200)) timeout 300 python bench.py $F --exchange $ex > $O/$ex.log 2>&1
done
python - <<'PY' | tee $O/summary.txt
import json
for name in ("plain", "peer", "all_gather", "all_reduce"):
    for l in open(f"gpurun_out/r05x/{name}.log"):
        if l.startswith("{"):
            b = json.loads(l)
            print(f"{name:11s} ms_per_step {b['ms_per_step']:.5f}  cold {b['ms_per_step_cold']}  exchange: {b['config'].get('exchange')}")
PY
