"""Round 4: every number DESIGN.md / README.md / BASELINE.md quote for this round is generated from the committed profiles/r04_* files by this script and pasted between
the `<!-- r04:NAME:begin -->` / `<!-- r04:NAME:end -->` markers of those documents (no hand-typed ranges).  Usage: python scripts/r04_numbers.py [--write]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
ALG = 56_028_980  # 48 N + 16 buckets + 52 voxels + 560 for the headline


def jl(name):
    path = os.path.join(P, name)
    return [json.loads(l) for l in open(path) if l.startswith("{")] if os.path.exists(path) else []


def stats(name, needle):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return None
    for r in csv.DictReader(open(path)):
        if needle in r["Name"]:
            return dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3)
    return None


def frac(us):
    return ALG / (us * 1e-6) / 8e12


def rng(vals, fmt="{:.2f}"):
    vals = sorted(vals)
    return fmt.format(vals[0]) if len(vals) == 1 or fmt.format(vals[0]) == fmt.format(vals[-1]) else fmt.format(vals[0]) + " – " + fmt.format(vals[-1])


def headline():
    b = jl("r04_bench_n1.json")[-1]
    r = b["roofline"]
    split = json.load(open(os.path.join(P, "r04_kernel_trace_split.json")))
    sw = jl("r04_sweep.jsonl")
    by = {}
    for x in sw:
        by.setdefault(x["case"], []).append(x)
    traffic = json.load(open(os.path.join(P, "r04_hbm_traffic.json")))
    st = stats("r04_bench_kernel_stats.csv", "vgicp_stream_kernel")
    st_nw = stats("r04_bench_kernel_stats_no_warmup.csv", "vgicp_stream_kernel")
    st_nm = stats("r04_bench_kernel_stats_no_mirror.csv", "vgicp_stream_kernel")
    out = []
    out.append("| what (C2 headline: 1 M source points vs the 2 M-point map at 0.5 m; algorithmic bytes 56.03 MB) | µs | fraction of 8 TB/s on algorithmic bytes | source |")
    out.append("|---|---|---|---|")
    out.append(f"| **streaming part of the fused kernel inside the bench's timed steps** (first workgroup started → last partial row in; the kernel's own 100 MHz stamps) — `roofline.frac` | "
               f"**{r['kernel_ms']*1e3:.2f}** | **{r['frac']:.3f}** | `profiles/r04_bench_n1.json` |")
    out.append(f"| whole fused kernel in step (→ last part's sums on their way to the host) — `frac_fused_kernel` | {r['fused_kernel_ms']*1e3:.2f} | {r['frac_fused_kernel']:.3f} | same |")
    out.append(f"| tile kernel back to back (HIP events on the launch stream, two-kernel form) | {r['kernel_ms_back_to_back']*1e3:.2f} | {r['frac_back_to_back']:.3f} | same |")
    out.append(f"| on the bytes the launch really requests ({r['actual_bytes']/1e6:.2f} MB: 36 B per point through the packed mirror + block grid + records) — `frac_actual` | "
               f"{r['kernel_ms']*1e3:.2f} | {r['frac_actual']:.3f} | same |")
    for key, label in [("in_step", "rocprofv3 per dispatch: tile kernel inside a synchronous step of the two-kernel form (dispatch → end signal)"),
                       ("back_to_back", "rocprofv3 per dispatch: tile kernel back to back"),
                       ("fused_in_step", "rocprofv3 per dispatch: fused kernel inside a step (dispatch → end signal: includes the ~1 µs before the first workgroup starts and the "
                                          "end-of-kernel flush behind the host-memory stores)")]:
        if key in split:
            x = split[key]
            out.append(f"| {label} | {x['mean_us']:.2f} (median {x['median_us']:.2f}, n = {x['n']}) | {frac(x['mean_us']):.3f} | `profiles/r04_kernel_trace_split.txt` |")
    if st:
        out.append(f"| rocprofv3 `--stats` average of the driver's command, `bench.py --steps 20 --warmup 5` ({st['calls']} dispatches: the untimed device wake-up's fused steps dominate) | "
                   f"{st['avg_us']:.2f} | {frac(st['avg_us']):.3f} | `profiles/r04_bench_kernel_stats.csv` |")
    if st_nw:
        out.append(f"| the same command with `--device-warmup-ms 0` ({st_nw['calls']} dispatches, the mix of round 3's file: 25 fused steps + 101 back-to-back launches) | "
                   f"{st_nw['avg_us']:.2f} | {frac(st_nw['avg_us']):.3f} | `profiles/r04_bench_kernel_stats_no_warmup.csv` |")
    if st_nm:
        out.append(f"| the driver's command with `--no-mirror` (the caller's 12 + 36 B per point, round 3's stream) | {st_nm['avg_us']:.2f} | {frac(st_nm['avg_us']):.3f} | "
                   f"`profiles/r04_bench_kernel_stats_no_mirror.csv` |")

    def sw_row(case, label):
        if case in by:
            xs = by[case]
            out.append(f"| sweep, in step: {label} | {rng([x['stream_us'] for x in xs])} (step {rng([x['step_us'] for x in xs], '{:.1f}')}; back to back {rng([x['b2b_us'] for x in xs])}) | "
                       f"{rng([x['frac_in_step'] for x in xs], '{:.3f}')} | `profiles/r04_sweep.jsonl` |")
    sw_row("1:-1:1024", "library defaults (mirror, XCD share table, skew 150)")
    sw_row("0:-1:1024", "without the packed mirror")
    sw_row("1:-1:1024:0:0", "with equal XCD shares")
    sw_row("1:250:1024", "with round 3's skew (250)")
    sw_row("1:0:1024", "flat split")
    out.append(f"| fabric traffic per launch (PMC FETCH_SIZE × {traffic['fetch_scale']:.3f} calibrated + WRITE_SIZE) | {traffic['tile_kernel_hbm_bytes_per_launch']/1e6:.2f} MB "
               f"({traffic['tile_kernel_hbm_bytes_per_launch']/r['actual_bytes']:.2f} × the requested bytes) | — | `profiles/r04_pmc_summary.txt` |")
    out.append("")
    out.append(f"Step host to host {b['ms_per_step']*1e3:.1f} µs = {b['value']:.3g} point-correspondences/s; parity against the reference's own CPU code "
               f"{max(v for k, v in b['parity_vs_oracle'].items() if k != 'num_inliers_equal'):.1e} (gate 1e-5); the reference's CPU factor on the same box "
               f"{b['cpu_baseline']['ms_per_linearize']:.0f} ms with {b['cpu_baseline']['cores']} threads ({b['cpu_baseline']['ms_per_linearize_1thread']:.0f} ms with one).")
    return "\n".join(out)


def results():
    b = jl("r04_bench_n1.json")[-1]
    c, big, c4 = b["configs"], b.get("big_source"), b["c4"]
    out = ["| config | points / call | host → host ms | throughput | dominant kernel, roofline fraction | parity (max rel, H / b / error) | CPU (reference code) |", "|---|---|---|---|---|---|---|"]
    r = b["roofline"]
    par = max(v for k, v in b["parity_vs_oracle"].items() if k != "num_inliers_equal")
    out.append(f"| **C2 headline**: 1 factor, 1 M pts vs 2 M-pt map @0.5 m | 1.0 M | **{b['ms_per_step']:.4f}** | **{b['value']:.3g} corr/s** | streaming part in step {r['kernel_ms']*1e3:.2f} µs = "
               f"**{r['frac']:.3f}** of 8 TB/s (whole fused kernel {r['frac_fused_kernel']:.3f}; back to back {r['frac_back_to_back']:.3f}) | {par:.1e} | "
               f"{b['cpu_baseline']['ms_per_linearize']:.0f} ms @{b['cpu_baseline']['cores']} thr |")
    if big and "roofline" in big:
        out.append(f"| the same factor with an 8 M-point source (beyond the Infinity Cache) | 8.0 M | {big['ms_per_linearize']:.4f} | {big['value']:.3g} corr/s | in step {big['roofline']['kernel_ms']*1e3:.1f} µs = "
                   f"**{big['roofline']['frac']:.3f}** (back to back {big['roofline']['frac_back_to_back']:.3f}) | — | — |")
    x = c["C1"]
    out.append(f"| C1: two full kitti_00 scans, k = 10 covariances, 0.5 m | {x['points']/1e3:.1f} k | {x['ms']:.4f} | {x['corr_per_s']:.3g} corr/s | {x['roofline']['kernel_ms']*1e3:.1f} µs back to back, "
               f"{x['roofline']['frac']:.2f} (launch-bound) | {max(v for k, v in x['parity_vs_reference'].items() if k != 'num_inliers_equal'):.1e} | {x['cpu_baseline']['ms']:.2f} ms |")
    x = c["C3"]
    out.append(f"| C3: 256-factor submap graph, 1.0 m, ONE batched call | {x['points']/1e6:.2f} M | {x['ms']:.4f} (with copy {x['ms_with_copy']:.4f}) | {x['corr_per_s']:.3g} corr/s | "
               f"{x['roofline']['kernel_ms']*1e3:.1f} µs, {x['roofline']['frac']:.2f} algorithmic (re-reads hit L2: not an HBM fraction) | "
               f"{x['parity_vs_reference_max']:.1e} ({x['parity_factors_checked']} factors) | {x['cpu_baseline']['ms']:.0f} ms |")
    out.append(f"| C4: 4096 factors (whole job on ONE GPU through `ShardedLinearizer`; exchange `{c4['exchange']}`) | {c4['points_per_linearize']/1e6:.0f} M | {c4['ms_per_linearize']:.4f} | "
               f"{c4['value']:.3g} corr/s | tile kernel {c4['tile_kernel_ms_slowest_rank']:.3f} ms, {c4['algorithmic_frac_per_gpu']:.2f} algorithmic (not an HBM fraction) | tests | — |")
    x = c["C5"]
    out.append(f"| C5: k-NN covariances (k = 10), 1 M pts | 1.0 M | {x['covariances']['ms']:.4f} | {x['covariances']['points_per_s']:.3g} pts/s | not HBM-bound (DESIGN §4.8) | median "
               f"{x['covariances']['parity_vs_reference']['rel_err_median']:.1e} | {x['covariances']['cpu_baseline']['ms']:.0f} ms |")
    out.append(f"| C5: GICP linearise, 1 M vs 1 M pts | 1.0 M | {x['gicp']['ms']:.4f} | {x['gicp']['corr_per_s']:.3g} corr/s | not HBM-bound | "
               f"{max(v for k, v in x['gicp']['parity_vs_reference'].items() if k != 'num_inliers_equal'):.1e} | {x['gicp']['cpu_baseline']['ms']:.0f} ms |")
    x = c["map_build"]
    out.append(f"| voxel-map build, 2 M pts @0.5 m | 2.0 M | {x['ms']:.4f} | {x['points_per_s']:.3g} pts/s | whole call {x['roofline']['frac']:.3f} of 8 TB/s on its 96 MB (launch- and latency-bound) | "
               f"bit-reproducible; = reference CPU map through save/load | — |")
    return "\n".join(out)


def misc():
    out = []
    warm = jl("r04_warm.jsonl")
    if warm:
        cold = [x["stream_us"] for x in warm if x["block"].startswith("cold")][:2]
        settled = [x["stream_us"] for x in warm if "2000-step" in x["block"]]
        idle = [x["stream_us"] for x in warm if "after 2 s idle" in x["block"]][:2]
        out.append(f"* device wake-up (`profiles/r04_warm.jsonl`): streaming part {rng(cold)} µs in the first two 200-step blocks behind 2 s of idle, {rng(idle)} µs again after another 2 s of idle, "
                   f"{rng(settled)} µs settled (2000-step blocks).")
    s8 = jl("r04_sweep_8m.jsonl")
    if s8:
        d = [x for x in s8 if x["case"] == "1:-1:1024"]
        nm = [x for x in s8 if x["case"] == "0:-1:1024"]
        out.append(f"* 8 M-point source (`profiles/r04_sweep_8m.jsonl`): in step {rng([x['stream_us'] for x in d], '{:.1f}')} µs = {rng([x['frac_in_step'] for x in d], '{:.3f}')} with the library "
                   f"defaults, {rng([x['stream_us'] for x in nm], '{:.1f}')} µs = {rng([x['frac_in_step'] for x in nm], '{:.3f}')} without the mirror.")
    mb = jl("r04_map_build.json")
    if mb:
        out.append(f"* map build (`profiles/r04_map_build.json`, `r04_map_build_kernel_stats.csv`): {mb[-1]['map_build_ms_median']:.3f} ms per 2 M points (round 3: 0.49); "
                   f"k-NN covariances {mb[-1]['covariances_ms_median']:.3f} ms per 1 M points (round 3: 1.04–1.10).")
    c5 = jl("r04_c5_staging.jsonl")
    if c5:
        d0 = [x["ms_median"] for x in c5 if x["structure"] == 0]
        d6 = [x["ms_median"] for x in c5 if x["structure"] == 6]
        if d0 and d6:
            out.append(f"* covariance queries heavy-first vs plain cell-sorted order (`profiles/r04_c5_staging.jsonl`, structures 0 / 6): {rng(d0, '{:.3f}')} vs {rng(d6, '{:.3f}')} ms per call.")
    return "\n".join(out)


SECTIONS = {"headline": headline, "results": results, "misc": misc}

if __name__ == "__main__":
    texts = {k: f() for k, f in SECTIONS.items()}
    for k, t in texts.items():
        print(f"<!-- r04:{k} -->\n{t}\n")
    if "--write" in sys.argv:
        for doc in ["DESIGN.md", "README.md", "BASELINE.md"]:
            path = os.path.join(ROOT, doc)
            s = open(path).read()
            for k, t in texts.items():
                pat = re.compile(r"(<!-- r04:%s:begin -->\n).*?(<!-- r04:%s:end -->)" % (k, k), re.S)
                s = pat.sub(lambda m: m.group(1) + t + "\n" + m.group(2), s)
            open(path, "w").write(s)
