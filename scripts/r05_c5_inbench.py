"""Round 5, C5: where the time of one gp_estimate_covariances call goes when it runs INSIDE bench.py (configs.C5.covariances.ms reads ~0.07 ms above scripts/r05_c5.py on
the same box).  Input: the output directory of `rocprofv3 --kernel-trace --hip-trace --output-format csv` around either program.  For the last calls on 1 M-point clouds:
every kernel of the call (hardware queue, stream, start, duration) and every HIP API call of the launching thread inside the call's window (start, duration).
Usage: python scripts/r05_c5_inbench.py <rocprofv3 output dir> [calls]"""
import csv, glob, os, sys

root = sys.argv[1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def find(pattern):
    hits = sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True), key=os.path.getsize)
    return hits[-1] if hits else None


kt, at = find("*kernel_trace.csv"), find("*hip_api_trace.csv")
if not kt:
    sys.exit(f"no kernel trace under {root}")
kern = []
with open(kt) as f:
    for r in csv.DictReader(f):
        kern.append(dict(name=r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gp::", "").replace("(anonymous namespace)::", ""), q=r["Queue_Id"], s=r.get("Stream_Id", "?"),
                         tid=r["Thread_Id"], t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"]), grid=int(r["Grid_Size_X"]), scratch=r.get("Scratch_Size", "?"),
                         vgpr=r.get("VGPR_Count", "?")))
kern.sort(key=lambda k: k["t0"])
api = []
if at:
    with open(at) as f:
        for r in csv.DictReader(f):
            api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"]))
    api.sort()
far = [i for i, k in enumerate(kern) if k["name"].startswith("covariance_far_kernel")]
calls = []
for i in far:
    j = i
    while j >= 0 and not kern[j]["name"].startswith("bins_bbox_kernel"):
        j -= 1
    if j < 0:
        continue
    t_begin = kern[j]["t0"]
    ks = [k for k in kern[j:] if k["t0"] <= kern[i]["t1"] + 2_000_000 and k["t0"] >= t_begin]
    # the call ends with its last covariance kernel
    last_cov = max((k["t1"] for k in ks if k["name"].startswith("covariance") and k["t0"] < kern[i]["t1"] + 1000), default=kern[i]["t1"])
    ks = [k for k in ks if k["t0"] < last_cov]
    calls.append(dict(points=max(k["grid"] for k in ks if k["name"].startswith("covariance_kernel")), t_begin=t_begin, t_end=last_cov, kernels=ks, tid=kern[i]["tid"]))
big = [c for c in calls if c["points"] > 900_000]
print(f"{kt}: {len(kern)} dispatches, {len(far)} covariance calls ({len(big)} on ~1 M points); queues used by the process: {sorted(set(k['q'] for k in kern))}")
spans = [round((c["t_end"] - c["t_begin"]) / 1e3, 1) for c in big]
print("first kernel -> last covariance kernel of every 1 M-point call, us:", spans)
for c in big[-want:]:
    print(f"\n== call on {c['points']} query threads; kernels (queue / stream / scratch B / VGPRs) ==")
    for k in c["kernels"]:
        print(f"  +{(k['t0'] - c['t_begin']) / 1e3:8.1f} us  dur {(k['t1'] - k['t0']) / 1e3:7.1f}  q {k['q']} s {k['s']}  scratch {k['scratch']} vgpr {k['vgpr']}  grid {k['grid']:>8}  {k['name'][:40]}")
    if not api:
        continue
    lo, hi = c["t_begin"] - 400_000, c["t_end"] + 300_000
    inside = [a for a in api if lo <= a[0] <= hi]
    tids = {}
    for a in inside:
        tids[a[3]] = tids.get(a[3], 0) + 1
    main = max(tids, key=tids.get) if tids else None
    print(f"  HIP API calls of thread {main} from 400 us before the first kernel to 300 us after the last ({len(inside)} calls; those of 3 us or more, and every allocation / event / wait):")
    total = {}
    for a in inside:
        if a[3] != main:
            continue
        d = (a[1] - a[0]) / 1e3
        total.setdefault(a[2], [0, 0.0])
        total[a[2]][0] += 1
        total[a[2]][1] += d
        if d >= 3.0 or any(w in a[2] for w in ("Malloc", "Free", "Event", "Synchronize", "Wait")):
            print(f"    +{(a[0] - c['t_begin']) / 1e3:8.1f} us  {d:7.1f} us  {a[2]}")
    print("  totals:", ", ".join(f"{k} x{v[0]} {v[1]:.1f} us" for k, v in sorted(total.items(), key=lambda kv: -kv[1][1])))
