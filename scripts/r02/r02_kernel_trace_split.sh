#!/bin/bash
# rocprofv3 per-dispatch durations of the tile kernel in the bench command, split by launch pattern: inside a synchronous step (the previous
# dispatch on the queue is a finalize kernel) vs back to back with itself (the roofline loop of bench.py).
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-c4 --kernel-iters 100 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/r02_kernel_trace_split.txt
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = {"in a synchronous step (previous dispatch: finalize kernel)": [], "back to back (previous dispatch: tile kernel)": []}
prev = None
for r in rows:
    name = r["Kernel_Name"]
    if "vgicp_pipeline" in name:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if prev is not None and "vgicp_pipeline" in prev["Kernel_Name"]:
            groups["back to back (previous dispatch: tile kernel)"].append((d, (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3))
        elif prev is not None and "finalize" in prev["Kernel_Name"]:
            groups["in a synchronous step (previous dispatch: finalize kernel)"].append((d, (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3))
    prev = r
for k, v in groups.items():
    if not v:
        continue
    d = np.array([x[0] for x in v]); g = np.array([x[1] for x in v])
    print(f"{k}: n={len(d)}  duration us: mean {d.mean():.2f} median {np.median(d):.2f} min {d.min():.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f};  gap to the previous dispatch us: median {np.median(g):.2f}")
PY
