"""Round 5, C5: when does the second covariance launch (side stream) start relative to the first (caller's stream), per call, from the kernel rows scripts/r05_trace_tail.py cut out
of rocprofv3 --kernel-trace runs of bench.py.  Usage: python scripts/r05_c5_queue_pipes.py <label>=<kernels.csv> ...   (writes the covariance rows next to the output as CSV when
--keep <dir> is given)"""
import csv, os, sys

keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
for arg in [a for a in sys.argv[1:] if "=" in a]:
    label, path = arg.split("=", 1)
    rows = [r for r in csv.DictReader(open(path)) if "covariance" in r["Kernel_Name"]]
    for r in rows:
        r["t0"], r["t1"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["t0"])
    if keep:
        with open(os.path.join(keep, f"r05_c5_queue_pipes_{label}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Queue_Id", "Stream_Id", "Kernel_Name", "Grid_Size_X", "Start_Timestamp", "End_Timestamp"])
            for r in rows:
                w.writerow([r["Queue_Id"], r["Stream_Id"], r["Kernel_Name"].split("(")[0].replace("void gp::", ""), r["Grid_Size_X"], r["t0"], r["t1"]])
    cov = [r for r in rows if "covariance_kernel" in r["Kernel_Name"]]
    far = [r for r in rows if "covariance_far" in r["Kernel_Name"]]
    by_size = {}
    for a, b in zip(cov, cov[1:]):
        if a["Grid_Size_X"] != b["Grid_Size_X"] or a["Queue_Id"] == b["Queue_Id"] or b["t0"] >= a["t1"]:  # (the second launch of a call starts while the first runs)
            continue
        f = next((x for x in far if x["t0"] >= a["t1"] - 1000 and x["t0"] < a["t1"] + 50_000), None)
        end = max(b["t1"], f["t1"] if f else 0)
        by_size.setdefault(int(a["Grid_Size_X"]), []).append(((b["t0"] - a["t0"]) / 1e3, (a["t1"] - a["t0"]) / 1e3, (b["t1"] - b["t0"]) / 1e3, (end - a["t0"]) / 1e3, a["Queue_Id"], b["Queue_Id"]))
    print(f"{label}: {len(cov) // 2} calls")
    for size, v in sorted(by_size.items()):
        med = lambda i: sorted(x[i] for x in v)[len(v) // 2]
        print(f"  {size:>8} query threads, {len(v):2d} calls: caller's queue {v[0][4]}, side stream's queue {v[0][5]} | second launch starts {min(x[0] for x in v):6.1f} .. {max(x[0] for x in v):6.1f} us after the first "
              f"(median {med(0):6.1f}) | first launch {med(1):5.0f} us, second {med(2):5.0f} us | first start -> all three kernels done {med(3):5.0f} us (median)")
        print("           per call in time order (bench.py: target cloud, source, source, ...), start of the second launch / all done, us: " + "  ".join(f"{x[0]:.0f}/{x[3]:.0f}" for x in v))
