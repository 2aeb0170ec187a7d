#!/bin/bash
# Round 6: SQ counters of the headline's stream kernel (7 fused steps of bench.py --pmc-child): where do the waves' cycles go?
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06; mkdir -p $O; R=$PWD
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/ssq$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/ssq$i -o p -- python3 $R/bench.py --pmc-child > /dev/null 2> $O/stream_sq_$i.err)
  f=$(find /tmp/ssq$i -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|vgicp_stream_kernel" $f > $O/stream_sq_$i.csv
done
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O = sys.argv[1]
v = defaultdict(list)
for p in sorted(glob.glob(os.path.join(O, "stream_sq_*.csv"))):
    for r in csv.DictReader(open(p)):
        v[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in v.items()}
out = ["# vgicp_stream_kernel (headline: 1 M points, fused), mean per launch over the fused steps of bench.py --pmc-child; SQ_* count quad-cycles summed over waves"]
for k in sorted(m):
    out.append(f"{k:24s} {m[k]:16.1f}  (n={len(v[k])})")
wc = m.get("SQ_WAVE_CYCLES", 0)
if wc:
    out.append(f"shares of the waves' cycles: VALU active {m.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}  any instruction active {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}  issue-stalled {m.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  parked (s_waitcnt / barrier) {m.get('SQ_WAIT_ANY', 0) / wc:.3f}  LDS issue-stalled {m.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}")
    out.append(f"VALU wave-instructions per launch {m.get('SQ_INSTS_VALU', 0):.0f} = {m.get('SQ_INSTS_VALU', 0) * 4 / (1024 * 2.4e9) * 1e6:.2f} us of issue at 2.4 GHz on 1024 SIMDs; waves {m.get('SQ_WAVES', 0):.0f}; mean wave lifetime {wc * 4 / max(m.get('SQ_WAVES', 1), 1) / 2.1e3:.2f} us at 2.1 GHz")
open(os.path.join(O, "stream_sq.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
