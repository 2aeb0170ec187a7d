#!/bin/bash
# fabric traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the tile kernel on the batched configs C3 and C4-shard
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; : > $O/r02_configs_traffic.txt
cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt && timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pt -o p -- python scripts/bench_configs.py C3,C4 > /tmp/pt.log 2>&1
  f=$(find /tmp/pt -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $ctr >> $O/r02_configs_traffic.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "vgicp_pipeline" in row["Kernel_Name"] and row["Counter_Name"] == sys.argv[2]:
        acc[int(row["Grid_Size"])].append(float(row["Counter_Value"]))
for g, v in sorted(acc.items()):
    print(f"{sys.argv[2]:10s} tile kernel, grid {g:8d} threads ({g // 256} workgroups): mean {sum(v)/len(v):12.1f} KiB per launch (n={len(v)})")
PY
done
cat $O/r02_configs_traffic.txt
