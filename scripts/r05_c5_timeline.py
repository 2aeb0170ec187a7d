"""Round 5, C5: start / end of the covariance kernels of the last call per cloud out of a rocprofv3 --kernel-trace of scripts/r05_c5.py (the two launches of covariance_kernel
on two streams and covariance_far_kernel behind the first).  Usage: python scripts/r05_c5_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "covariance" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-9:]
names = ["c5_source", "c5_target", "kitti_00"]
for c in range(3):
    trio = last[3 * c: 3 * c + 3]
    t0 = int(trio[0]["Start_Timestamp"])
    print(names[c], "| " + " | ".join(f"{r['Kernel_Name'].split('(')[0].replace('void gp::', '')[:32]} {round((int(r['Start_Timestamp']) - t0) / 1e3, 1)} .. {round((int(r['End_Timestamp']) - t0) / 1e3, 1)} us" for r in trio),
          "| all done at", round(max(int(r["End_Timestamp"]) for r in trio) / 1e3 - t0 / 1e3, 1), "us")
