"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) into per-kernel means and a calibrated
HBM-traffic figure for the tile kernel.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE under-reports streaming reads (MI355X_MICROARCH.md,
HBM section), so the read side is scaled by k = known_bytes / FETCH_SIZE measured on calibration_stream_kernel, which
reads exactly 48 * n bytes with the tile kernel's own access pattern.

usage: pmc_summary.py <fetch_csv> <write_csv> <n_source_points> <out_json>
"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, ctr):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == ctr:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def find(d, needle):
    for k, v in d.items():
        if needle in k:
            return v
    return None


fetch_csv, write_csv, n_src, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
fetch, nf = per_kernel(fetch_csv, "FETCH_SIZE")
write, _ = per_kernel(write_csv, "WRITE_SIZE")
for k in sorted(fetch, key=lambda k: -fetch[k]):
    print(f"FETCH_SIZE[KiB] mean={fetch[k]:12.1f} launches={nf[k]:4d} WRITE_SIZE[KiB] mean={write.get(k, float('nan')):10.1f}  {k[:100]}")
calib = find(fetch, "calibration_stream_kernel")
tile_f = find(fetch, "vgicp_stream") or find(fetch, "vgicp_pipeline")
tile_w = find(write, "vgicp_stream") or find(write, "vgicp_pipeline") or 0.0
res = {"fetch_size_kib": tile_f, "write_size_kib": tile_w, "calibration_fetch_kib": calib, "known_calibration_bytes": 48 * n_src}
if calib and tile_f:
    k = 48.0 * n_src / (calib * 1024.0)
    res["fetch_scale"] = k
    res["tile_kernel_hbm_bytes_per_launch"] = int(tile_f * 1024.0 * k + tile_w * 1024.0)
print(json.dumps(res))
json.dump(res, open(out, "w"), indent=1)
