"""groups the per-dispatch counter rows of vgicp_stream_kernel from a rocprofv3 --pmc run of scripts/r03_c4_traffic.py (PMC=1) by configuration
(dispatch order, REPS launches each) and prints the means per launch"""
import csv
import sys
from collections import defaultdict

path, configs, reps = sys.argv[1], sys.argv[2].split(","), int(sys.argv[3])
rows = defaultdict(dict)
for r in csv.DictReader(open(path)):
    if "vgicp_stream_kernel" in r["Kernel_Name"]:
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
assert len(ids) == len(configs) * reps, (len(ids), len(configs), reps)
for k, cfg in enumerate(configs):
    grp = [rows[i] for i in ids[k * reps + 1 : (k + 1) * reps]]  # (the first launch of a configuration follows a table rebuild: left out)
    names = sorted(grp[0])
    print(cfg, " ".join(f"{n}={sum(g[n] for g in grp) / len(grp):.1f}" for n in names))
