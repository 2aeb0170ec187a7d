#!/bin/bash
# Round-4 evidence run on the GPU box (through gpurun).  Everything lands under gpurun_out/r04/ and is copied into profiles/r04_* afterwards (scripts/r04_collect.sh).
# PMC passes are separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r04; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt; nproc >> $O/device.txt
B="--no-cpu-baseline --no-c4 --no-configs"
# 1. per-kernel times of the bench command (driver's form: --steps 20 --warmup 5; the step runs the fused finalize on the packed mirror), with the per-dispatch trace for the
#    launch-pattern split; the same with --finalize two-kernel (the tile kernel alone inside a step) and with --no-mirror (the caller's 48 B per point)
rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 20 --warmup 5 $B > $O/rocprof_bench.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -4 $f | cut -c1-200
tf=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
rm -rf /tmp/prof2 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o bench -- python bench.py --steps 20 --warmup 5 $B --finalize two-kernel > $O/rocprof_bench_two_kernel.log 2>&1
f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats_two_kernel.csv && head -3 $f | cut -c1-200
t=$(find /tmp/prof2 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python scripts/kernel_trace_split.py "$t" $O/kernel_trace_split.json "$tf" | tee $O/kernel_trace_split.txt
rm -rf /tmp/prof3 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o bench -- python bench.py --steps 20 --warmup 5 $B --no-mirror > $O/rocprof_bench_no_mirror.log 2>&1
f=$(find /tmp/prof3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats_no_mirror.csv && head -3 $f | cut -c1-200
# (the mix of round 3's file: 25 fused steps + the back-to-back launches, without the untimed device wake-up's ~10^4 fused steps)
rm -rf /tmp/prof4 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o bench -- python bench.py --steps 20 --warmup 5 $B --device-warmup-ms 0 > $O/rocprof_bench_no_warmup.log 2>&1
f=$(find /tmp/prof4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats_no_warmup.csv && head -3 $f | cut -c1-200
# 2. HBM traffic of the tile kernel: FETCH_SIZE / WRITE_SIZE in separate passes, read side calibrated on a known stream (48 B per point, the API layout's pattern)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  GP_BENCH_CALIBRATE=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o pmc -- python bench.py --steps 5 --warmup 2 $B --kernel-iters 5 --device-warmup-ms 0 > $O/pmc_$ctr.log 2>&1
done
ff=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_summary.py "$ff" "$fw" 1000000 $O/hbm_traffic.json > $O/pmc_summary.txt; tail -1 $O/pmc_summary.txt | cut -c1-300
# 3. SQ / TCC counters of the tile kernel
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pc && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc -o p -- python bench.py --steps 5 --warmup 2 $B --kernel-iters 5 --device-warmup-ms 0 > /tmp/pc.log 2>&1
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
# 4. the round's sweeps and probes
timeout 300 python scripts/r04_sweep.py "1:-1:1024,0:-1:1024,1:-1:1024:0:0,1:0:1024,1:100:1024,1:150:1024,1:250:1024,1:-1:768,1:-1:1024:1" --steps 400 2>/dev/null > $O/sweep.jsonl; cut -c1-215 $O/sweep.jsonl | head -9
timeout 300 python scripts/r04_sweep.py "1:-1:1024,1:100:1024,1:250:1024,1:-1:1024:0:0,0:-1:1024" --points 8000000 --steps 200 2>/dev/null > $O/sweep_8m.jsonl; cut -c1-215 $O/sweep_8m.jsonl | head -5
timeout 200 python scripts/r04_instep_xcd.py 2>/dev/null | grep "^{" > $O/instep_xcd.jsonl
timeout 200 python scripts/r04_instep_xcd.py 8=1000 9=1000 10=1000 11=1000 12=1000 13=1000 14=1000 15=1000 2>/dev/null | grep "^{" > $O/instep_xcd_equal_shares.jsonl; head -1 $O/instep_xcd.jsonl $O/instep_xcd_equal_shares.jsonl
timeout 100 python scripts/r04_warm.py 2>/dev/null | grep "^{" > $O/warm.jsonl; head -3 $O/warm.jsonl
timeout 60 ./scripts/probe/dispatch_ramp_probe > $O/dispatch_ramp_probe.txt 2>&1; head -3 $O/dispatch_ramp_probe.txt
# 5. C5 and the map build
timeout 200 python scripts/r04_c5.py 0,6 2>/dev/null | grep "^{" > $O/c5_staging.jsonl; head -4 $O/c5_staging.jsonl
[ -f gtsam_points_amd/libgtsam_points_hip_wavelog.so ] && timeout 200 python scripts/r04_c5_wavelog.py 0 2>&1 | grep -v amdgpu > $O/c5_wavelog.txt
rm -rf /tmp/pm && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o mb -- python scripts/r04_map_build.py > $O/map_build.log 2>&1
f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/map_build_kernel_stats.csv
t=$(find /tmp/pm -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python scripts/r04_build_timeline.py $t > $O/build_timeline.txt
timeout 60 ./scripts/probe/sort_probe > $O/sort_probe.txt 2>&1; timeout 60 ./scripts/probe/bins_probe > $O/bins_probe.txt 2>&1; tail -2 $O/sort_probe.txt | cut -c1-200
timeout 100 python scripts/r04_map_build.py 2>/dev/null | grep "^{" > $O/map_build.json; cat $O/map_build.json
# 6. the whole default bench (driver's clock), smoke and the GPU test-suite
timeout 900 python bench.py > $O/bench.log 2>&1; grep "^{" $O/bench.log > $O/bench_n1.json; cut -c1-300 $O/bench_n1.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1800 python -m pytest -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
ls -la $O
