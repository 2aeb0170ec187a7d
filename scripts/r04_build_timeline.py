"""Per-dispatch timeline of the last voxel-map build and the last k-NN covariance call in a rocprofv3 kernel trace of scripts/r04_map_build.py (start offset, duration, gap)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def show(rows, first, last, title):
    idx_last = max(i for i, r in enumerate(rows) if last in r["Kernel_Name"])
    idx_first = max(i for i, r in enumerate(rows[: idx_last + 1]) if first in r["Kernel_Name"])
    while idx_first > 0 and "fillBuffer" in rows[idx_first - 1]["Kernel_Name"]:
        idx_first -= 1
    t0 = int(rows[idx_first]["Start_Timestamp"]); prev_end = t0
    print(title)
    for r in rows[idx_first : idx_last + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{(s - t0) / 1e3:8.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8s}  {r['Kernel_Name'][:70]}")
        prev_end = e
    print(f"  total {(prev_end - t0) / 1e3:.2f} us")
show(rows, "bins_bbox_kernel", "insert_voxels_kernel", "voxel-map build (last)")
show(rows, "bins_bbox_kernel", "covariance_kernel", "k-NN covariances (last)")
