#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; : > $O/r02_idle_gap.txt; cd $GRAFT_REPO_ROOT
for spin in ${SPINS:-0}; do
echo "=== spinner of $spin us on a second stream before every pass ===" | tee -a $O/r02_idle_gap.txt
rm -rf /tmp/ig && GP_PROBE_SPIN_US=$spin timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ig -o ig -- python scripts/probe/idle_gap_probe.py > /tmp/ig.log 2>&1
f=$(find /tmp/ig -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee -a $O/r02_idle_gap.txt
import csv, sys
import numpy as np
rows = sorted((r for r in csv.DictReader(open(sys.argv[1])) if "spin_kernel" not in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
pts = []
prev = None
for r in rows:
    if "vgicp_pipeline" in r["Kernel_Name"] and prev is not None:
        pts.append(((int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    prev = r
pts = np.array(pts)
print("tile kernel duration by idle gap in front of it (rocprofv3 dispatch timestamps, C2):")
for lo, hi in [(0, 8), (8, 12), (12, 16), (16, 25), (25, 40), (40, 80), (80, 400), (400, 5000), (5000, 1e9)]:
    m = (pts[:, 0] >= lo) & (pts[:, 0] < hi)
    if m.sum():
        print(f"  gap {lo:6.0f} .. {hi:8.0f} us: n={int(m.sum()):4d}  duration mean {pts[m, 1].mean():6.2f} median {np.median(pts[m, 1]):6.2f} min {pts[m, 1].min():6.2f} us")
PY
done
