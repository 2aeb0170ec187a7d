#!/bin/bash
# usage: pmc_tile.sh VARIANT  -> gpurun_out/pmc_tile_vVARIANT.txt (per-kernel means of several counter passes)
V=${1:-1}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_tile_v$V.txt; : > $OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  case "$line" in \#*) continue;; esac
  if [ -n "${PMC_QUICK:-}" ]; then case "$line" in SQ_WAVES*|SQ_INSTS_VALU*|TCC_HIT*) ;; *) continue;; esac; fi
  i=$((i+1)); D=gpurun_out/pmc_tile_tmp_$i; rm -rf $D
  timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $D -o p -- python scripts/profile_tile.py $V 10 > $D.log 2>&1
  f=$(find $D -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" >> $OUT <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "vgicp_pipeline" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:40s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
  else echo "pass '$line' failed: $(tail -2 $D.log)" >> $OUT; fi
  rm -rf $D
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE
TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
LIST
cat $OUT
