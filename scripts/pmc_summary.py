"""Summarise a rocprofv3 counter_collection.csv: per-kernel mean counter value."""
import csv
import sys
from collections import defaultdict

path, ctr = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") == ctr:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{ctr} kernel={k[:90]} launches={len(v)} mean={sum(v)/len(v):.1f} total={sum(v):.1f}")
