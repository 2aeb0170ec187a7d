"""Summary of scripts/r06/c5_pmc.sh: per covariance launch (kernel instantiation x grid size: gp_estimate_covariances issues the heavy cells' queries, the rest, and the far
queries as three launches) the mean counter values, its rocprofv3 --kernel-trace duration, and the fractions of the stated bounds.  Writes c5_pmc.json beside the text."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

O = sys.argv[1]
SIMDS, CLOCK_HZ, XCDS = 1024, 2.4e9, 8  # 256 CUs x 4 SIMDs; MI355X peak engine clock (MI355X_MICROARCH.md)


def key_of(name, grid):
    short = name.split("(")[0].replace("void ", "").replace("gp::", "")
    return f"{short} grid {grid}"


vals = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(O, "c5_pmc_pass*.csv"))):
    for row in csv.DictReader(open(path)):
        if "covariance" in row["Kernel_Name"]:
            vals[key_of(row["Kernel_Name"], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = defaultdict(list)
trace = os.path.join(O, "c5_kernel_trace.csv")
if os.path.exists(trace):
    for row in csv.DictReader(open(trace)):
        grid = row.get("Grid_Size") or str(int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1) * int(row.get("Grid_Size_Z", 1) or 1))
        dur[key_of(row["Kernel_Name"], grid)].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3)
run = json.load(open(os.path.join(O, "c5_run.json"))) if os.path.exists(os.path.join(O, "c5_run.json")) else {}
out = dict(run=run, launches={})
print("# C5 covariance launches, 1 M-point cloud (BASELINE configs[4]); counters = mean per launch (rocprofv3 --pmc, separate passes, source cloud only);")
print("# duration = rocprofv3 --kernel-trace per dispatch, mean after the first three calls")
print("# run (un-profiled, alternating clouds as bench.py does):", json.dumps(run))
for name in sorted(vals):
    m = {k: sum(v) / len(v) for k, v in vals[name].items()}
    n = {k: len(v) for k, v in vals[name].items()}
    for k in sorted(m):
        print(f"{name:44s} {k:26s} mean/launch {m[k]:16.1f}  (n={n[k]})")
    o = dict(counters=m)
    d = dur.get(name, [])
    d = d[3:] if len(d) > 4 else d
    if d and "SQ_INSTS_VALU" in m:
        us = sum(d) / len(d)
        # issue bound: every VALU wave-instruction occupies its SIMD's issue port for one quad-cycle (4 clocks; f64 FMA is full rate on CDNA4, transcendentals are 4x)
        t_issue_us = (m["SQ_INSTS_VALU"] + 3.0 * m.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) * 4.0 / (SIMDS * CLOCK_HZ) * 1e6
        clock = m["GRBM_GUI_ACTIVE"] / XCDS / (us * 1e-6) if "GRBM_GUI_ACTIVE" in m else None
        o.update(duration_us=round(us, 2), valu_issue_bound_us=round(t_issue_us, 2), frac_valu_issue=round(t_issue_us / us, 4))
        print(f"{name:44s} {'duration':26s} mean/launch {us:16.2f} us  (n={len(d)})")
        print(f"{name:44s} {'VALU issue bound':26s} {t_issue_us:10.2f} us at {CLOCK_HZ / 1e9:.1f} GHz -> frac {t_issue_us / us:.3f}" + (f"   (clock of the counter pass: GRBM_GUI_ACTIVE / 8 XCDs / duration = {clock / 1e9:.2f} GHz)" if clock else ""))
        if "SQ_WAVE_CYCLES" in m:
            shares = dict(valu_active=m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], issue_stalled=m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], parked=m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"])
            o["wave_cycle_shares"] = {k: round(v, 4) for k, v in shares.items()}
            print(f"{name:44s} {'wave-cycle shares':26s} VALU active {shares['valu_active']:.3f}  issue-stalled {shares['issue_stalled']:.3f}  parked (s_waitcnt / barrier) {shares['parked']:.3f}")
        if "FETCH_SIZE" in m:
            mb = m["FETCH_SIZE"] * 1024 * 2 / 1e6
            o.update(hbm_fetch_mb=round(mb, 2), frac_hbm=round(mb * 1e6 / (us * 1e-6) / 8e12, 5))
            print(f"{name:44s} {'HBM fetch':26s} {mb:10.2f} MB per launch (FETCH_SIZE KiB x 2: the gfx950 correction) -> {mb * 1e6 / (us * 1e-6) / 8e12:.4f} of 8 TB/s")
    out["launches"][name] = o
json.dump(out, open(os.path.join(O, "c5_pmc.json"), "w"), indent=1)
