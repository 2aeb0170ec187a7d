#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; rm -rf $O; mkdir -p $O
for mode in 0 1; do
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pc && GP_COV_FAST=$mode timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc -o p -- python scripts/r03_c5.py > /tmp/pc.log 2>&1
  f=$(find /tmp/pc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $mode >> $O/c5_pmc.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Kernel_Name"]
    if "covariance_" in n:
        acc[(n.split("(")[0][-34:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"fast={sys.argv[2]} {k:36s} {c:30s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
done
cat $O/c5_pmc.txt
