#!/bin/bash
# SQ counters of the structure-build kernels (separate rocprofv3 run, --kernel-trace only)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/build_check; mkdir -p $O
rm -rf /tmp/pc && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pc -o p -- python scripts/r04_map_build.py > /tmp/pc.log 2>&1
f=$(find /tmp/pc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $O/build_pmc.txt
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join(f"{c}={sum(v)/len(v):.0f}" for c, v in sorted(d.items())))
PY
