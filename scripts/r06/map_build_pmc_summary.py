"""Per kernel of the voxel-map build: HBM bytes per build (rocprofv3 --pmc FETCH_SIZE x 2 -- the gfx950 correction of MI355X_MICROARCH.md's HBM section -- + WRITE_SIZE, KiB),
duration per build, and the totals against the build's algorithmic bytes (48 B per point read once).  Writes map_build_pmc.json."""
import csv
import json
import os
import sys
from collections import defaultdict

O = sys.argv[1]
BUILDS_PMC, BUILDS_STATS = 12, 25


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("gp::", "")


tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(float)
    path = os.path.join(O, f"map_build_pmc_{ctr}.csv")
    if os.path.exists(path):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == ctr:
                acc[short(row["Kernel_Name"])] += float(row["Counter_Value"])
    tot[ctr] = acc
dur = {}
path = os.path.join(O, "map_build_kernel_stats.csv")
if os.path.exists(path):
    for row in csv.DictReader(open(path)):
        dur[short(row["Name"])] = dur.get(short(row["Name"]), 0.0) + float(row["TotalDurationNs"])
run = json.load(open(os.path.join(O, "map_build_run.json"))) if os.path.exists(os.path.join(O, "map_build_run.json")) else {}
BUILD_KERNELS = ["bins_bbox_kernel", "bins_key_kernel", "radix_onesweep_kernel", "bins_count_kernel", "bins_cells_kernel", "segmented_stats_kernel", "insert_voxels_kernel", "host_flag_kernel"]
out = dict(run=run, kernels={})
print("# voxel-map build, 2 M points @0.5 m: per kernel and build -- HBM read (FETCH_SIZE KiB x 2 x 1024), written (WRITE_SIZE KiB x 1024), rocprofv3 duration")
print("# run (un-profiled):", json.dumps(run))
sum_r = sum_w = sum_us = 0.0
for k in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]) | set(dur)):
    if not any(k.startswith(b) for b in BUILD_KERNELS):
        continue
    r = tot["FETCH_SIZE"].get(k, 0.0) * 2048.0 / BUILDS_PMC
    w = tot["WRITE_SIZE"].get(k, 0.0) * 1024.0 / BUILDS_PMC
    us = dur.get(k, 0.0) / BUILDS_STATS * 1e-3
    sum_r, sum_w, sum_us = sum_r + r, sum_w + w, sum_us + us
    out["kernels"][k] = dict(read_mb=round(r / 1e6, 2), written_mb=round(w / 1e6, 2), us=round(us, 2))
    print(f"{k:42s} read {r / 1e6:8.2f} MB  written {w / 1e6:8.2f} MB  {us:8.2f} us")
alg = 48.0 * run.get("points", 2_000_000)
out.update(traffic_bytes_per_build=int(sum_r + sum_w), read_bytes=int(sum_r), written_bytes=int(sum_w), kernel_us_per_build=round(sum_us, 2), algorithmic_bytes=int(alg))
print(f"{'TOTAL per build':42s} read {sum_r / 1e6:8.2f} MB  written {sum_w / 1e6:8.2f} MB  {sum_us:8.2f} us of kernels; algorithmic {alg / 1e6:.1f} MB -> traffic / algorithmic = {(sum_r + sum_w) / alg:.2f}")
json.dump(out, open(os.path.join(O, "map_build_pmc.json"), "w"), indent=1)
