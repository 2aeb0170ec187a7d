#!/bin/bash
# C5 with the straight-line insertion (TopK<KMAX, true>) and the square-root-free filter bound: tests first (identical neighbour sets), then wall time, rocprofv3 kernel time and SQ counters; GP_COV_FULL=0 for the A/B
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_knn_gicp_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest exit $?" >> $O/pytest.txt; tail -4 $O/pytest.txt | cut -c1-300
timeout 300 python scripts/r03_c5.py 2>&1 | grep "^{" | tee -a $O/c5.jsonl
GP_COV_FULL=0 timeout 300 python scripts/r03_c5.py 2>&1 | grep "^{" | tee -a $O/c5.jsonl
rm -rf /tmp/pk && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python scripts/r03_c5.py > /tmp/c5_prof.log 2>&1
f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_kernel_stats.csv && head -6 $f | cut -c1-220
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pc && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pc -o p -- python scripts/r03_c5.py > /tmp/pc.log 2>&1
  f=$(find /tmp/pc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $O/c5_pmc.txt <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "covariance_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in sorted(acc.items()):
    print(f"covariance_kernel {c:28s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
cat $O/c5_pmc.txt
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "c5" > $O/pytest_c5.txt 2>&1; echo "pytest exit $?" >> $O/pytest_c5.txt; tail -3 $O/pytest_c5.txt | cut -c1-300
