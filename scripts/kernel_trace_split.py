"""rocprofv3 per-dispatch durations of the tile kernel in a bench.py run, split by launch pattern: inside a synchronous step (the previous dispatch on
the queue is a finalize kernel -- or, in the fused form, the previous step's tile kernel -- and the queue stood empty while the host collected the
result) vs back to back (the roofline loops of bench.py).
usage: kernel_trace_split.py <kernel_trace.csv of `bench.py --finalize two-kernel`> <out.json> [<kernel_trace.csv of the default (fused) run>]"""
import csv
import json
import sys

import numpy as np

TILE = ("vgicp_stream_kernel", "vgicp_pipeline")


def classify(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    groups = {"in_step": [], "back_to_back": [], "fused_in_step": []}
    prev = None
    for r in rows:
        name = r["Kernel_Name"]
        if any(t in name for t in TILE):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3 if prev is not None else 0.0
            if prev is not None and any(t in prev["Kernel_Name"] for t in TILE):
                groups["fused_in_step" if gap > 3.0 else "back_to_back"].append((d, gap))  # fused form: one launch per step, the host in between
            elif prev is not None and "finalize" in prev["Kernel_Name"] and gap > 3.0:
                groups["in_step"].append((d, gap))
            elif prev is not None and "finalize" in prev["Kernel_Name"]:
                groups["back_to_back"].append((d, gap))  # the whole-pass loop of gp_vgicp_batch_time_linearize: tile, finalize, tile, ... without the host in between
        prev = r
    return groups


def summarise(groups, label):
    out = {}
    for k, v in groups.items():
        if not v:
            continue
        d = np.array([x[0] for x in v])
        g = np.array([x[1] for x in v])
        out[k] = dict(n=int(len(d)), mean_us=round(float(d.mean()), 3), median_us=round(float(np.median(d)), 3), min_us=round(float(d.min()), 3), p90_us=round(float(np.percentile(d, 90)), 3),
                      max_us=round(float(d.max()), 3), gap_median_us=round(float(np.median(g)), 2))
        print(f"[{label}] {k}: n={len(d)}  duration us: mean {d.mean():.2f} median {np.median(d):.2f} min {d.min():.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f};  gap to the previous dispatch us: median {np.median(g):.2f}")
    alld = np.array([x[0] for v in groups.values() for x in v])
    if len(alld):
        out["all"] = dict(n=int(len(alld)), mean_us=round(float(alld.mean()), 3))
        print(f"[{label}] all tile-kernel dispatches: n={len(alld)} mean {alld.mean():.2f} us")
    return out


out = summarise(classify(sys.argv[1]), "two-kernel form")
if len(sys.argv) > 3:
    fused = summarise(classify(sys.argv[3]), "fused form (default)")
    if "fused_in_step" in fused:
        out["fused_in_step"] = fused["fused_in_step"]
    out["all_fused_run"] = fused.get("all")
json.dump(out, open(sys.argv[2], "w"), indent=1)
