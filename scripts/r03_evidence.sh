#!/bin/bash
# Round-3 evidence run on the GPU box (through gpurun).  Everything lands under gpurun_out/r03/ and is copied into profiles/r03_* afterwards.
# PMC passes are separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r03; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt; nproc >> $O/device.txt
# 1. per-kernel times of the bench command (driver's form: --steps 20 --warmup 5; the step runs the fused finalize), with the per-dispatch trace for the
#    launch-pattern split; the same with --finalize two-kernel (the tile kernel alone inside a step)
rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-configs > $O/rocprof_bench.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -5 $f | cut -c1-200
tf=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
rm -rf /tmp/prof2 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 --no-configs --finalize two-kernel > $O/rocprof_bench_two_kernel.log 2>&1
f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats_two_kernel.csv && head -5 $f | cut -c1-200
t=$(find /tmp/prof2 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python scripts/kernel_trace_split.py "$t" $O/kernel_trace_split.json "$tf" | tee $O/kernel_trace_split.txt
# 2. HBM traffic of the tile kernel: FETCH_SIZE / WRITE_SIZE in separate passes, read side calibrated on a known stream
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  GP_BENCH_CALIBRATE=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c4 --no-configs --kernel-iters 5 > $O/pmc_$ctr.log 2>&1
done
ff=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_summary.py "$ff" "$fw" 1000000 $O/hbm_traffic.json > $O/pmc_summary.txt; tail -1 $O/pmc_summary.txt | cut -c1-300
# 3. SQ counters of the tile kernel
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pc && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c4 --no-configs --kernel-iters 5 > /tmp/pc.log 2>&1
  f=$(find /tmp/pc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $O/pmc_tile_sq.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "vgicp_stream_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"vgicp_stream_kernel {c:34s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cat $O/pmc_tile_sq.txt
# 4. C4 shard: times per schedule, then the traffic counters per schedule
CFG="0:0:0,1:0:0,0:2:0,0:0:8,1:0:8,1:0:16"
C4_CONFIGS=$CFG timeout 600 python scripts/r03_c4_traffic.py > $O/c4_times.jsonl 2> $O/c4_times.err; cut -c1-260 $O/c4_times.jsonl
rocprofv3 --list-avail 2>/dev/null | grep -E "TCC_EA0_RDREQ|TCC_EA_RDREQ|FETCH_SIZE|TCC_BUBBLE|TCC_EA0_RD_UNCACHED" | head -12 > $O/avail_counters.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/pc4 && PMC=1 C4_CONFIGS=$CFG timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc4 -o p -- python scripts/r03_c4_traffic.py > /tmp/pc4.log 2>&1
  f=$(find /tmp/pc4 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/r03_c4_traffic_parse.py "$f" $CFG 4 >> $O/c4_traffic.txt 2>> $O/c4_traffic.err
done
cat $O/c4_traffic.txt; tail -3 $O/c4_traffic.err 2>/dev/null
# 4b. fused finalize A/B and tail timeline, surface validation at size, block-sparse solver, C5 kernel stats, synchronous batched calls under rocprofv3
timeout 300 python scripts/r03_fused.py 1500 4 > $O/fused_finalize.jsonl 2>/dev/null; cut -c1-260 $O/fused_finalize.jsonl | head -6
timeout 200 python scripts/r03_sv_time.py 2>/dev/null | grep "^{" > $O/surface_validation_time.jsonl; cat $O/surface_validation_time.jsonl
timeout 600 python scripts/solver_time.py > $O/solver_time.jsonl 2>/dev/null; python - <<'PY'
import json, os
for l in open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r03/solver_time.jsonl"):
    d = json.loads(l)
    print(d["graph"], d["poses"], {k.replace("sparse_", "").replace("_solve_ms", ""): v for k, v in d.items() if k.endswith("_solve_ms") and k.startswith("sparse")})
PY
rm -rf /tmp/pk && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python scripts/r03_c5.py > $O/c5.log 2>&1
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv && head -3 $f | cut -c1-200; grep "^{" $O/c5.log
for wl in c4 c3; do
  rm -rf /tmp/pb && PMC=1 C4_CONFIGS=0:0:0 WORKLOAD=$wl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python scripts/r03_c4_traffic.py > /tmp/pb.log 2>&1
  f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "vgicp_stream_kernel|finalize" $f | cut -c1-60,170-260 | tee -a $O/sync_kernels_$wl.txt
done
# 5. the whole default bench (driver's clock) and the GPU test-suite, smoke
timeout 900 python bench.py > $O/bench.log 2>&1; grep "^{" $O/bench.log > $O/bench_n1.json; cut -c1-400 $O/bench_n1.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1800 python -m pytest -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
ls -la $O
