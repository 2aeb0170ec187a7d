"""Round 5: every number DESIGN.md / README.md / BASELINE.md quote for this round is generated from the committed profiles/r05_* files by this script and pasted between
the `<!-- r05:NAME:begin -->` / `<!-- r05:NAME:end -->` markers of those documents (no hand-typed figures; tests/test_docs_cpu.py regenerates and compares).
Usage: python scripts/r05_numbers.py [--write]"""
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
    vals = sorted(v for v in vals if v is not None)
    if not vals:
        return "—"
    return fmt.format(vals[0]) if len(vals) == 1 or fmt.format(vals[0]) == fmt.format(vals[-1]) else fmt.format(vals[0]) + " – " + fmt.format(vals[-1])


def runs():
    return [jl(f"r05_bench_run{i}.json")[-1] for i in (1, 2, 3) if jl(f"r05_bench_run{i}.json")]


def headline():
    rs = runs()
    b = rs[0]
    r = b["roofline"]
    split = json.load(open(os.path.join(P, "r05_kernel_trace_split.json")))
    st = stats("r05_bench_kernel_stats.csv", "vgicp_stream_kernel")
    st_nw = stats("r05_bench_kernel_stats_no_warmup.csv", "vgicp_stream_kernel")
    ex = jl("r05_kernel_experiments.jsonl")
    out = []
    out.append("| what (C2 headline: 1 M source points vs the 2 M-point map at 0.5 m; algorithmic bytes 56.03 MB) | µs | fraction of 8 TB/s on algorithmic bytes | source |")
    out.append("|---|---|---|---|")
    out.append(f"| **the WHOLE fused kernel inside the driver command's timed steps** (first workgroup started → last part's sums on their way to the host; the kernel's own 100 MHz stamps) — `roofline.frac` | "
               f"**{rng([x['roofline']['kernel_ms'] * 1e3 for x in rs])}** | **{rng([x['roofline']['frac'] for x in rs], '{:.3f}')}** | `profiles/r05_bench_run{{1,2,3}}.json` (three runs of `bench.py --gpus 1 --steps 20 --warmup 5` on one box) |")
    out.append(f"| its streaming part (→ last partial row in: rounds 3–4's `frac`) — `frac_streaming` | {rng([x['roofline']['streaming_ms'] * 1e3 for x in rs])} | {rng([x['roofline']['frac_streaming'] for x in rs], '{:.3f}')} | same |")
    out.append(f"| the same K steps BEFORE the untimed device wake-up (`--device-warmup-ms 0`, rounds 1–3's protocol) — `roofline.cold` | {rng([x['roofline']['cold']['fused_kernel_us'] for x in rs])} "
               f"(streaming {rng([x['roofline']['cold']['stream_us'] for x in rs])}) | {rng([frac(x['roofline']['cold']['fused_kernel_us']) for x in rs], '{:.3f}')} | same |")
    out.append(f"| tile kernel back to back (HIP events on the launch stream, two-kernel form) | {rng([x['roofline']['kernel_ms_back_to_back'] * 1e3 for x in rs])} | {rng([x['roofline']['frac_back_to_back'] for x in rs], '{:.3f}')} | same |")
    out.append(f"| on the bytes the launch really requests ({r['actual_bytes'] / 1e6:.2f} MB: 36 B per point through the packed mirror + block grid + records) — `frac_actual` | "
               f"{rng([x['roofline']['kernel_ms'] * 1e3 for x in rs])} | {rng([x['roofline']['frac_actual'] for x in rs], '{:.3f}')} | same |")
    out.append(f"| fabric traffic per launch, measured IN THE RUN (two `rocprofv3 --pmc` passes over a child run; FETCH_SIZE × {r['traffic_detail']['fetch_scale']:.3f} calibrated + WRITE_SIZE) — `roofline.traffic` | "
               f"{rng([x['roofline']['traffic'] / 1e6 for x in rs])} MB ({rng([x['roofline']['traffic'] / x['roofline']['actual_bytes'] for x in rs])} × the requested bytes) | — | same |")
    for key, label in [("fused_in_step", "rocprofv3 per dispatch: fused kernel inside a step (dispatch → end signal: includes the ~1 µs before the first workgroup starts and the end-of-kernel flush behind the host-memory stores)"),
                       ("in_step", "rocprofv3 per dispatch: tile kernel inside a synchronous step of the two-kernel form"),
                       ("back_to_back", "rocprofv3 per dispatch: tile kernel back to back")]:
        if key in split:
            s = split[key]
            out.append(f"| {label} | {s['mean_us']:.2f} (median {s['median_us']:.2f}, n = {s['n']}) | {frac(s['mean_us']):.3f} | `profiles/r05_kernel_trace_split.txt` |")
    if st:
        out.append(f"| rocprofv3 `--stats` average of the driver's command (`--steps 20 --warmup 5`, {st['calls']} dispatches: the untimed wake-up's fused steps dominate) | {st['avg_us']:.2f} | {frac(st['avg_us']):.3f} | `profiles/r05_bench_kernel_stats.csv` |")
    if st_nw:
        out.append(f"| the same command with `--device-warmup-ms 0` ({st_nw['calls']} dispatches) | {st_nw['avg_us']:.2f} | {frac(st_nw['avg_us']):.3f} | `profiles/r05_bench_kernel_stats_no_warmup.csv` |")
    for e, label in [(0, "A/B of the bounded attempt (VERDICT r04 #1), in step: the product kernel"), (1, "… with the block grid warmed behind the first request (#1b)"),
                     (2, "… with R C_A Rᵀ in f32 (#1c upper bound; breaks parity, timing only)")]:
        rows = [x for x in ex if x["experiment"] == e]
        if rows:
            out.append(f"| {label} | {rng([x['fused_us'] for x in rows])} (streaming {rng([x['stream_us'] for x in rows])}; step {rng([x['step_us'] for x in rows], '{:.1f}')}) | "
                       f"{rng([x['frac_whole_kernel'] for x in rows], '{:.3f}')} | `profiles/r05_kernel_experiments.jsonl` |")
    p = b.get("parity_vs_oracle") or {}
    worst = max(v for k, v in p.items() if k != "num_inliers_equal") if p else float("nan")
    c = b["cpu_baseline"]
    out.append("")
    out.append(f"Step host to host {rng([x['ms_per_step'] * 1e3 for x in rs], '{:.1f}')} µs (before the wake-up: {rng([x['ms_per_step_cold'] * 1e3 for x in rs], '{:.1f}')}) = {rng([x['value'] for x in rs], '{:.3g}')} point-correspondences/s; "
               f"parity against the reference's own CPU code {worst:.1e} (gate 1e-5); the reference's CPU factor on the same box {c['ms_per_linearize']:.0f} ms with {c['cores']} threads "
               f"({c['ms_per_linearize_1thread']:.0f} ms with one).  The driver's round-4 run: 21.85 µs per step, whole kernel 12.84 µs = 0.546 (streaming slice 0.613), rocprofv3 `--stats` 14.51 µs = 0.483.")
    return "\n".join(out)


def results():
    rs = runs()
    b = rs[0]
    r, cfg = b["roofline"], b["configs"]
    big, c4 = b.get("big_source") or {}, b["c4"]
    p = b.get("parity_vs_oracle") or {}
    worst = max(v for k, v in p.items() if k != "num_inliers_equal") if p else float("nan")
    c1, c3, c5, mb = cfg["C1"], cfg["C3"], cfg["C5"], cfg["map_build"]
    pc1 = max(v for k, v in c1["parity_vs_reference"].items() if k != "num_inliers_equal")
    pg = max(v for k, v in c5["gicp"]["parity_vs_reference"].items() if k != "num_inliers_equal")
    out = ["| config | points / call | host → host ms | throughput | dominant kernel, roofline fraction | parity (max rel, H / b / error) | CPU (reference code) |", "|---|---|---|---|---|---|---|"]
    out.append(f"| **C2 headline**: 1 factor, 1 M pts vs 2 M-pt map @0.5 m | 1.0 M | **{rng([x['ms_per_step'] for x in rs], '{:.4f}')}** | **{rng([x['value'] for x in rs], '{:.3g}')} corr/s** | "
               f"WHOLE fused kernel in step {rng([x['roofline']['kernel_ms'] * 1e3 for x in rs])} µs = **{rng([x['roofline']['frac'] for x in rs], '{:.3f}')}** of 8 TB/s (streaming slice {rng([x['roofline']['frac_streaming'] for x in rs], '{:.3f}')}; back to back {rng([x['roofline']['frac_back_to_back'] for x in rs], '{:.3f}')}) | {worst:.1e} | "
               f"{b['cpu_baseline']['ms_per_linearize']:.0f} ms @{b['cpu_baseline']['cores']} thr |")
    if big.get("roofline"):
        br = big["roofline"]
        out.append(f"| the same factor with an 8 M-point source (beyond the Infinity Cache) | 8.0 M | {big['ms_per_linearize']:.4f} | {big['value']:.3g} corr/s | streaming part in step {br['kernel_ms'] * 1e3:.1f} µs = **{br['frac']:.3f}** (back to back {br['frac_back_to_back']:.3f}) | — | — |")
    out.append(f"| C1: two full kitti_00 scans, k = 10 covariances, 0.5 m | {c1['points'] / 1e3:.1f} k | {c1['ms']:.4f} | {c1['corr_per_s']:.3g} corr/s | {c1['roofline']['kernel_ms'] * 1e3:.1f} µs back to back, {c1['roofline']['frac']:.2f} (launch-bound) | {pc1:.1e} | "
               f"{c1['cpu_baseline']['ms']:.2f} ms |")
    out.append(f"| C3: 256-factor submap graph, 1.0 m, ONE batched call | {c3['points'] / 1e6:.2f} M | {c3['ms']:.4f} (with copy {c3['ms_with_copy']:.4f}) | {c3['corr_per_s']:.3g} corr/s | {c3['roofline']['kernel_ms'] * 1e3:.1f} µs, {c3['roofline']['frac']:.2f} algorithmic "
               f"(re-reads hit L2: not an HBM fraction) | {c3['parity_vs_reference_max']:.1e} ({c3['parity_factors_checked']} factors) | {c3['cpu_baseline']['ms']:.0f} ms |")
    out.append(f"| C4: 4096 factors (whole job on ONE GPU through `ShardedLinearizer`; exchange `{c4['exchange']}`) | {c4['points_per_linearize'] / 1e6:.0f} M | {c4['ms_per_linearize']:.4f} | {c4['value']:.3g} corr/s | "
               f"tile kernel {c4['tile_kernel_ms_slowest_rank']:.3f} ms, {c4['algorithmic_frac_per_gpu']:.2f} algorithmic (not an HBM fraction) | tests | — |")
    cov = c5["covariances"]
    out.append(f"| C5: k-NN covariances (k = 10), 1 M pts (the config's cloud / the map-like sampling of the scene / a real kitti_00 scan: `profiles/r05_c5_ab.jsonl`) | 1.0 M | {cov['ms']:.4f} / {cov['ms_target_cloud']:.4f} / "
               f"{cov.get('ms_kitti_scan', float('nan')):.4f} | {cov['points_per_s']:.3g} pts/s | not HBM-bound (DESIGN §4.8) | median {cov['parity_vs_reference']['rel_err_median']:.1e} | {cov['cpu_baseline']['ms']:.0f} ms |")
    out.append(f"| C5: GICP linearise, 1 M vs 1 M pts | 1.0 M | {c5['gicp']['ms']:.4f} | {c5['gicp']['corr_per_s']:.3g} corr/s | not HBM-bound | {pg:.1e} | {c5['gicp']['cpu_baseline']['ms']:.0f} ms |")
    out.append(f"| voxel-map build, 2 M pts @0.5 m | 2.0 M | {mb['ms']:.4f} | {mb['points_per_s']:.3g} pts/s | whole call {mb['roofline']['frac']:.3f} of 8 TB/s on its 96 MB (launch- and latency-bound) | bit-reproducible; = reference CPU map through save/load | — |")
    return "\n".join(out)


def lm():
    b = runs()[0]
    out = ["| graph | back end | iterations (inner) | ms per iteration, host to host | linearise | solve | error trials | harness glue | gate (0.015 rad / 0.15 m) |", "|---|---|---|---|---|---|---|---|---|"]
    for key, name in (("lm_c3", "C3: 256 factors / 64 submaps, pose 0 held, from ground truth ∘ Expmap(U(−0.1, 0.1)⁶)"), ("lm_c1", "C1: scan 000001 onto the map of scan 000000, from the identity")):
        o = b["configs"].get(key) or {}
        if "error" in o or not o:
            out.append(f"| {name} | — | — | {o.get('error', 'not run')} | | | | | |")
            continue
        for leg, label in (("gpu_device_solve", "GPU, records stay in HBM, damped build + block-sparse LLᵀ as ONE call (`gp_sparse_system_step` / `gp_dense_system_step`)"),
                           ("gpu_device_solve_three_calls", "… the same as round 4's three calls (build, download of b and c, solve: two waits, four copies)"), ("gpu_host_solve", "GPU linearise / error, numpy solve on the host (Python system builder: its time is under `linearise`)"),
                           ("cpu_baseline", f"the reference's CPU factor ({o['cpu_baseline']['cores']} threads) + numpy solve")):
            if leg not in o:
                continue
            x = o[leg]
            ph = x["ms_per_iteration_by_phase"]
            out.append(f"| {name} | {label} | {x['iterations']} ({x['inner_iterations']}) | **{x['ms_per_iteration']:.4f}** | {ph['linearize']:.4f} | {ph['solve']:.4f} | {ph['error']:.4f} | {ph['glue']:.4f} | "
                       f"{'met' if x['gate_met'] else 'NOT met'}: {x['max_rotation_error_rad']:.5f} rad / {x['max_translation_error_m']:.4f} m |")
    return "\n".join(out)


def c5():
    return open(os.path.join(P, "r05_c5_summary.txt")).read().rstrip("\n")


SECTIONS = {"headline": headline, "results": results, "lm": lm, "c5": c5}


def main():
    blocks = {k: f() for k, f in SECTIONS.items()}
    if "--write" not in sys.argv:
        for k, v in blocks.items():
            print(f"<!-- r05:{k}:begin -->\n{v}\n<!-- r05:{k}:end -->\n")
        return
    for doc in ("DESIGN.md", "README.md", "BASELINE.md"):
        path = os.path.join(ROOT, doc)
        text = open(path, encoding="utf-8").read()
        for k, v in blocks.items():
            text = re.sub(r"(<!-- r05:%s:begin -->\n).*?(\n<!-- r05:%s:end -->)" % (k, k), lambda m: m.group(1) + v + m.group(2), text, flags=re.S)
        open(path, "w", encoding="utf-8").write(text)
    print("written")


if __name__ == "__main__":
    main()
