"""Cut the C5 part out of a rocprofv3 --kernel-trace --hip-trace run of bench.py (or scripts/r05_c5.py): every kernel dispatch and HIP API call from 2 ms before the
N-th last covariance_far_kernel on, as two small CSV files.  Usage: python scripts/r05_trace_tail.py <rocprofv3 output dir> <out prefix> [N = 12]"""
import csv, glob, os, sys

root, prefix = sys.argv[1], sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 12


def find(pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True), key=os.path.getsize)
    return hits[-1] if hits else None


kt, at = find("*kernel_trace.csv"), find("*hip_api_trace.csv")
rows = list(csv.DictReader(open(kt)))
far = sorted(int(r["Start_Timestamp"]) for r in rows if "covariance_far_kernel" in r["Kernel_Name"])
t_lo = far[-min(nth, len(far))] - 2_000_000
t_hi = far[-1] + 3_000_000
keep = ["Queue_Id", "Stream_Id", "Thread_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Scratch_Size", "VGPR_Count", "Grid_Size_X", "Workgroup_Size_X"]
with open(prefix + "_kernels.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(keep)
    for r in rows:
        if t_lo <= int(r["Start_Timestamp"]) <= t_hi:
            w.writerow([r.get(k, "")[:60] if k == "Kernel_Name" else r.get(k, "") for k in keep])
if at:
    with open(prefix + "_api.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Function", "Thread_Id", "Start_Timestamp", "End_Timestamp"])
        for r in csv.DictReader(open(at)):
            if t_lo <= int(r["Start_Timestamp"]) <= t_hi:
                w.writerow([r["Function"], r["Thread_Id"], r["Start_Timestamp"], r["End_Timestamp"]])
print(prefix, "kernel rows from", t_lo, "to", t_hi)
