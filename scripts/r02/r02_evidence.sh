#!/bin/bash
# Round-2 evidence run on the GPU box (through gpurun).  Everything lands under gpurun_out/r02/ and is copied into profiles/r02_*
# by hand afterwards.  PMC passes are separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r02; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt; nproc >> $O/device.txt
timeout 1500 python -m pytest -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.log 2>&1; grep "^{" $O/bench.log > $O/bench_n1.json; cut -c1-300 $O/bench_n1.json
python scripts/trace_finalize.py 2>/dev/null | grep -E "^rep|^   " > $O/finalize_timeline.txt; tail -2 $O/finalize_timeline.txt
# per-kernel times of the same command
rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-c4 > $O/rocprof_bench.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -6 $f | cut -c1-160
# HBM traffic of the tile kernel: FETCH_SIZE / WRITE_SIZE in separate passes, read side calibrated on a known stream
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  GP_BENCH_CALIBRATE=1 timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-c4 --kernel-iters 5 > $O/pmc_$ctr.log 2>&1
done
ff=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python scripts/pmc_summary.py "$ff" "$fw" 1000000 $O/hbm_traffic.json > $O/pmc_summary.txt; tail -1 $O/pmc_summary.txt | cut -c1-300
# SQ / TCP / TCC counters of the default tile kernel
bash scripts/pmc_tile.sh 11 > /dev/null 2>&1; cp gpurun_out/pmc_tile_v11.txt $O/pmc_tile_sq.txt; head -12 $O/pmc_tile_sq.txt
# the other BASELINE configurations
timeout 1500 python scripts/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cut -c1-260 $O/configs.jsonl
# L2 hit rate of the tile kernel on the C4 shard
rm -rf /tmp/pc4 && timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/pc4 -o p -- python scripts/bench_configs.py C4 > /tmp/pc4.log 2>&1
f=$(find /tmp/pc4 -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" > $O/c4_tcc.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "vgicp_pipeline" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k, v in m.items():
    print(f"{k:16s} mean/launch {v:14.1f}  (n={len(acc[k])})")
if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
    print(f"TCC_HIT / (HIT + MISS) = {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}   (C4 shard: 512 factors x 32768 points, default tile kernel)")
PY
cat $O/c4_tcc.txt
# tile-kernel variants, per-workgroup timeline, 8 M-point source
timeout 900 python scripts/r02_sweep.py 1,2,3,4,8,9,10,11 0 --big > $O/sweep.jsonl 2> $O/sweep.err; cut -c1-200 $O/sweep.jsonl | tail -12
# C5: work counters, kernel times, PMC of the two kernels
timeout 300 python scripts/r02_profile_aux.py counters 2>/dev/null | grep "^{" > $O/c5_counters.jsonl; cat $O/c5_counters.jsonl
rm -rf /tmp/pk && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python scripts/r02_profile_aux.py c5 10 > $O/c5_prof.log 2>&1
grep "C5 1M" $O/c5_prof.log; f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pc && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc -o p -- python scripts/r02_profile_aux.py c5 3 > /tmp/pc.log 2>&1
  f=$(find /tmp/pc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $O/c5_pmc.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Kernel_Name"]
    if "covariance_kernel" in n or "gicp_tile_kernel" in n or "gicp_correspond_kernel" in n:
        acc[(n.split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:42s} {c:34s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cat $O/c5_pmc.txt | head -30
# voxel-map build
rm -rf /tmp/pm && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o map -- python scripts/r02_profile_aux.py map 10 > $O/map_prof.log 2>&1
grep "voxel map" $O/map_prof.log; f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/map_kernel_stats.csv
timeout 120 python scripts/r02_profile_aux.py map 20 2>/dev/null | grep "voxel map" > $O/map_wall.txt; cat $O/map_wall.txt
ls -la $O
