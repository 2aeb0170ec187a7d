#!/bin/bash
# The C5 part of scripts/r02_evidence.sh alone (after the GICP correspondence / algebra split): work counters, rocprofv3 kernel times, PMC,
# the bench_configs line and the fused-vs-split A/B.  Everything lands under gpurun_out/r02/.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O; rm -f $O/c5_pmc.txt
cd $GRAFT_REPO_ROOT
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
timeout 600 python scripts/bench_configs.py C5 > $O/configs_c5.jsonl 2> /dev/null; cut -c1-400 $O/configs_c5.jsonl
for sp in 0 1; do GP_GICP_SPLIT=$sp timeout 280 python scripts/gicp_split_ab.py 2>/dev/null | tail -1; done > $O/gicp_split_ab.jsonl; cat $O/gicp_split_ab.jsonl
