#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; rm -rf $O; mkdir -p $O
rm -rf /tmp/ps && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o s -- python scripts/r03_solver_prof.py 512 nd > $O/prof.log 2>&1
f=$(find /tmp/ps -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/solver_kernel_stats.csv && head -8 $f | cut -c1-220
grep "nnz" $O/prof.log
t=$(find /tmp/ps -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python - "$t" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last solve: take the last 30 kernels
last=rows[-16:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.2f} us  dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.2f} us  grid {r.get("Grid_Size_X", r.get("Grid_Size"))} wg {r.get("Workgroup_Size_X", r.get("Workgroup_Size"))}  {r["Kernel_Name"][:60]}')
PY
