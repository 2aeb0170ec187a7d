"""Round 6: every number DESIGN.md / README.md / BASELINE.md quote for this round is generated from the committed profiles/r06_* files by this script and pasted between
the `<!-- r06:NAME:begin -->` / `<!-- r06:NAME:end -->` markers of those documents (no hand-typed figures; tests/test_docs_cpu.py regenerates and compares).
Sources: profiles/r06_bench_run{1..6}.json = the LAST stdout line of six runs of the driver's command on two boxes (`python3 bench.py --gpus 1 --steps 20 --warmup 5`),
r06_bench_detail_run1.json = the detail file of run 1, r06_bench_kernel_stats*.csv = rocprofv3 --kernel-trace --stats of the headline protocol,
r06_bench_rehearsal_n2.json = the N = 2 launch on one GPU (gloo), r06_solver_step_time.jsonl, r06_wg_geometry*.jsonl / .csv, r06_c5_pmc.json, r06_map_build_pmc.json.
Usage: python scripts/r06_numbers.py [--write]"""
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


def jf(name):
    path = os.path.join(P, name)
    return json.load(open(path)) if os.path.exists(path) else {}


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
    return [jl(f"r06_bench_run{i}.json")[-1] for i in (1, 2, 3, 4, 5, 6) if jl(f"r06_bench_run{i}.json")]  # (1 - 3 and 4 - 6: two evidence calls = two boxes of the pool)


def headline():
    rs = runs()
    d = jf("r06_bench_detail_run1.json")
    st, st_nw, st_2k = (stats(f"r06_bench_kernel_stats{s}.csv", "vgicp_stream_kernel") for s in ("", "_no_warmup", "_two_kernel"))
    out = ["| what (C2 headline: 1 M source points vs the 2 M-point map at 0.5 m; algorithmic bytes 56.03 MB) | µs | fraction of 8 TB/s on algorithmic bytes | source |", "|---|---|---|---|"]
    out.append(f"| **the WHOLE fused kernel inside the driver command's timed steps** (first workgroup started → last part's sums on their way to the host; the kernel's own 100 MHz stamps) — `roofline.frac` | "
               f"**{rng([x['roofline']['kernel_ms'] * 1e3 for x in rs])}** | **{rng([x['roofline']['frac'] for x in rs], '{:.3f}')}** | `profiles/r06_bench_run{{1..6}}.json` (the last stdout line of six runs of `bench.py --gpus 1 --steps 20 --warmup 5`: runs 1 – 3 on one box of the pool, 4 – 6 on another; the slower box is 1 – 3) |")
    out.append(f"| its streaming part (→ last partial row in) — `frac_streaming` | {rng([x['roofline']['streaming_ms'] * 1e3 for x in rs])} | {rng([x['roofline']['frac_streaming'] for x in rs], '{:.3f}')} | same |")
    out.append(f"| **rocprofv3 `--kernel-trace --stats` average of the kernel, measured IN the run** (a child run of the headline protocol: wake-up + W + K steps; dispatch → end signal) — `roofline.rocprof_avg_ms`, `frac_rocprof` | "
               f"{rng([x['roofline']['rocprof_avg_ms'] * 1e3 for x in rs])} (n = {rng([x['roofline']['rocprof_calls'] for x in rs], '{:.0f}')}) | {rng([x['roofline']['frac_rocprof'] for x in rs], '{:.3f}')} | same |")
    if st:
        out.append(f"| the committed rocprofv3 `--stats` summary of the driver's command with the optional legs off (the 8 M-point leg runs the same instantiation on another workload) | {st['avg_us']:.2f} (n = {st['calls']}) | {frac(st['avg_us']):.3f} | `profiles/r06_bench_kernel_stats.csv` |")
    if st_nw:
        out.append(f"| … with `--device-warmup-ms 0` ({st_nw['calls']} dispatches: 25 + 25 fused steps + the back-to-back loop) | {st_nw['avg_us']:.2f} | {frac(st_nw['avg_us']):.3f} | `profiles/r06_bench_kernel_stats_no_warmup.csv` |")
    if st_2k:
        out.append(f"| … with `--finalize two-kernel` (the stream kernel without its fused tail, inside steps) | {st_2k['avg_us']:.2f} | {frac(st_2k['avg_us']):.3f} | `profiles/r06_bench_kernel_stats_two_kernel.csv` |")
    out.append(f"| stream kernel back to back (HIP events on the launch stream, two-kernel form) | {rng([x['roofline']['kernel_ms_back_to_back'] * 1e3 for x in rs])} | {rng([x['roofline']['frac_back_to_back'] for x in rs], '{:.3f}')} | `profiles/r06_bench_run{{1..6}}.json` |")
    out.append(f"| fabric traffic per launch, measured IN THE RUN (two `rocprofv3 --pmc` passes over a child run; FETCH_SIZE calibrated on a stream of known bytes + WRITE_SIZE) — `roofline.traffic` | "
               f"{rng([x['roofline']['traffic'] / 1e6 for x in rs])} MB ({rng([x['roofline']['traffic'] / ALG for x in rs])} × the algorithmic bytes: the 36-B packed mirror) | — | same |")
    wg = jl("r06_wg_geometry.jsonl")
    for w, label in ((4, "workgroup geometry A/B (VERDICT r05 #3), in step, alternating: 4 waves per workgroup (the product: 1024 workgroups)"), (16, "… 16 waves per workgroup (256 workgroups, one per compute unit)"),
                     (8, "… 8 waves per workgroup (512 workgroups)")):
        rows = [x for x in wg if x["wg_waves"] == w]
        if rows:
            ks = stats("r06_wg_geometry_kernel_stats.csv", f"false, 0, {w}>")
            out.append(f"| {label} | {rng([x['fused_us'] for x in rows])} (streaming {rng([x['stream_us'] for x in rows])}; step {rng([x['step_us'] for x in rows], '{:.1f}')}"
                       + (f"; rocprofv3 average {ks['avg_us']:.2f}" if ks else "") + f") | {rng([x['frac_whole_kernel'] for x in rows], '{:.3f}')} | `profiles/r06_wg_geometry.jsonl`, `r06_wg_geometry_kernel_stats.csv` |")
    c = rs[0]["cpu_baseline"]
    out.append("")
    out.append(f"Step host to host {rng([x['ms_per_step'] * 1e3 for x in rs], '{:.1f}')} µs (before the wake-up: {rng([x['ms_per_step_cold'] * 1e3 for x in rs], '{:.1f}')}) = {rng([x['value'] for x in rs], '{:.3g}')} point-correspondences/s; "
               f"parity against the reference's own CPU code {max(x['parity_max'] for x in rs):.1e} (gate 1e-5); the reference's CPU factor on the same box {rng([x['cpu_baseline']['ms_per_linearize'] for x in rs], '{:.0f}')} ms with "
               f"{c['cores']} threads ({rng([x['cpu_baseline']['ms_per_linearize_1thread'] for x in rs], '{:.0f}')} ms with one).  The line is {rng([len(json.dumps(x, separators=(',', ':'))) for x in rs], '{:.0f}')} bytes; "
               f"the whole run {rng([x['run_seconds'] for x in rs], '{:.0f}')} s (budget {d.get('budget_seconds', 45):.0f} s; legs skipped: {', '.join(rs[0]['legs_skipped']) or 'none'}).")
    return "\n".join(out)


def results():
    rs = runs()
    d = jf("r06_bench_detail_run1.json")
    cfg = d.get("configs") or {}
    big = d.get("big_source") or {}
    out = ["| config | points / call | host → host ms | throughput | dominant kernel, roofline fraction | parity (max rel, H / b / error) | CPU (reference code) |", "|---|---|---|---|---|---|---|"]
    out.append(f"| **C2 headline**: 1 factor, 1 M pts vs 2 M-pt map @0.5 m | 1.0 M | **{rng([x['ms_per_step'] for x in rs], '{:.4f}')}** | **{rng([x['value'] for x in rs], '{:.3g}')} corr/s** | "
               f"WHOLE fused kernel in step {rng([x['roofline']['kernel_ms'] * 1e3 for x in rs])} µs = **{rng([x['roofline']['frac'] for x in rs], '{:.3f}')}** of 8 TB/s by its own stamps, "
               f"**{rng([x['roofline']['frac_rocprof'] for x in rs], '{:.3f}')}** by rocprofv3's average (streaming slice {rng([x['roofline']['frac_streaming'] for x in rs], '{:.3f}')}; back to back {rng([x['roofline']['frac_back_to_back'] for x in rs], '{:.3f}')}) | "
               f"{max(x['parity_max'] for x in rs):.1e} | {rng([x['cpu_baseline']['ms_per_linearize'] for x in rs], '{:.0f}')} ms @{rs[0]['cpu_baseline']['cores']} thr |")
    if big.get("roofline"):
        br = big["roofline"]
        out.append(f"| the same factor with an 8 M-point source (beyond the Infinity Cache) | 8.0 M | {rng([x['legs'].get('big_source_ms') for x in rs], '{:.4f}')} | {big['value']:.3g} corr/s | streaming part in step {br['kernel_ms'] * 1e3:.1f} µs = "
                   f"**{rng([x['legs'].get('big_source_frac') for x in rs], '{:.3f}')}** (back to back {br['frac_back_to_back']:.3f}) | = headline kernel | — |")
    c1, c3, c5, mb = (cfg.get(k) or {} for k in ("C1", "C3", "C5", "map_build"))
    if c1.get("roofline"):
        pc1 = max(v for k, v in c1["parity_vs_reference"].items() if k != "num_inliers_equal")
        out.append(f"| C1: two full kitti_00 scans, k = 10 covariances, 0.5 m | {c1['points'] / 1e3:.1f} k | {rng([x['legs'].get('C1_ms') for x in rs], '{:.4f}')} | {c1['corr_per_s']:.3g} corr/s | {c1['roofline']['kernel_ms'] * 1e3:.1f} µs back to back, "
                   f"{c1['roofline']['frac']:.2f} (launch-bound) | {pc1:.1e} | {c1['cpu_baseline']['ms']:.2f} ms @{c1['cpu_baseline']['cores']} thr |")
    if c3.get("roofline"):
        out.append(f"| C3: 256-factor submap graph, 1.0 m, ONE batched call | {c3['points'] / 1e6:.2f} M | {rng([x['legs'].get('C3_ms') for x in rs], '{:.4f}')} (with copy {c3['ms_with_copy']:.4f}) | {c3['corr_per_s']:.3g} corr/s | {c3['roofline']['kernel_ms'] * 1e3:.1f} µs, "
                   f"{c3['roofline']['frac']:.2f} algorithmic (re-reads hit L2: not an HBM fraction) | {c3['parity_vs_reference_max']:.1e} ({c3['parity_factors_checked']} factors) | {rng([x['legs'].get('C3_cpu_ms') for x in rs], '{:.0f}')} ms (sampled) |")
    if c5.get("covariances"):
        cov = c5["covariances"]
        pg = max(v for k, v in c5["gicp"]["parity_vs_reference"].items() if k != "num_inliers_equal")
        out.append(f"| C5: k-NN covariances (k = 10), 1 M pts | 1.0 M | {rng([x['legs'].get('C5_cov_ms') for x in rs], '{:.4f}')} | {cov['points_per_s']:.3g} pts/s | VALU issue: {cov['roofline'].get('valu_wave_instructions_per_call', 0) / 1e6:.1f} M wave-instructions per call = "
                   f"**{rng([x['legs'].get('C5_cov_frac') for x in rs], '{:.3f}')}** of 614 G/s on the call's wall (`profiles/r06_c5_pmc.txt`) | median {cov['parity_vs_reference']['rel_err_median']:.1e} | {cov['cpu_baseline']['ms']:.0f} ms |")
        out.append(f"| C5: GICP linearise, 1 M vs 1 M pts | 1.0 M | {rng([x['legs'].get('C5_gicp_ms') for x in rs], '{:.4f}')} | {c5['gicp']['corr_per_s']:.3g} corr/s | not HBM-bound (DESIGN §4.8) | {pg:.1e} | {c5['gicp']['cpu_baseline']['ms']:.0f} ms |")
    if mb.get("roofline"):
        t = mb["roofline"].get("traffic")
        out.append(f"| voxel-map build, 2 M pts @0.5 m | 2.0 M | {rng([x['legs'].get('map_build_ms') for x in rs], '{:.4f}')} | {mb['points_per_s']:.3g} pts/s | whole call {rng([x['legs'].get('map_build_frac') for x in rs], '{:.3f}')} of 8 TB/s on its 96 MB; counter traffic "
                   + (f"{t / 1e6:.0f} MB per build = {mb['roofline'].get('frac_traffic', 0):.3f} (`profiles/r06_map_build_pmc.txt`)" if t else "—") + " | bit-reproducible; = reference CPU map through save/load | — |")
    fb = jf("r06_bench_detail_full_budget.json") if os.path.exists(os.path.join(ROOT, "profiles", "r06_bench_detail_full_budget.json")) else None
    if fb and fb.get("c4"):
        c4 = fb["c4"]
        out.append(f"| C4: {c4['factors']} factors over 512 submaps of 32,768 pts, 1.0 m, ONE batched call on ONE GPU (the leg the 45 s default budget skips: `bench.py --budget-seconds 120`, `profiles/r06_bench_full_budget.json`, whole run {fb.get('run_seconds', 0):.0f} s) | "
                   f"{c4['points_per_linearize'] / 1e6:.1f} M | {c4['ms_per_linearize']:.4f} | {c4['value']:.3g} corr/s | tile kernel {c4['tile_kernel_ms_slowest_rank'] * 1e3:.0f} µs; {c4['algorithmic_frac_per_gpu']:.2f} algorithmic (every source cloud serves 8 factors: not an HBM fraction) | = the C3 kernel | — |")
    reh = jl("r06_bench_rehearsal_n2.json")
    if reh:
        x = reh[-1]
        out.append(f"| N = 2 launch rehearsed on ONE GPU (`torch.distributed.run`, backend {x['backend']}: RCCL needs a device per rank) | 2 × 1.0 M | {x['ms_per_step']:.4f} | {x['value']:.3g} corr/s | exchange forms timed in the same job: "
                   + ", ".join(f"{k} {v:.4f} ms" if v is not None else f"{k} n/a" for k, v in x["exchange_ms"].items()) + f"; every form's stack verified bit for bit: {x['exchange_verified']}; c4 {x['legs'].get('c4_ms', float('nan')):.3f} ms, verified {x['legs'].get('c4_verified')} | "
                   f"rank 0's row {x['parity_max']:.1e} | — |")
    return "\n".join(out)


def lm():
    d = jf("r06_bench_detail_run1.json")
    rs = runs()
    out = ["| graph | back end | iterations (inner) | ms per iteration, host to host | linearise | solve | error trials | harness glue | gate (0.015 rad / 0.15 m) |", "|---|---|---|---|---|---|---|---|---|"]
    for key, name in (("lm_c3", "C3: 256 factors / 64 submaps, pose 0 held, from ground truth ∘ Expmap(U(−0.1, 0.1)⁶)"), ("lm_c1", "C1: scan 000001 onto the map of scan 000000, from the identity")):
        o = (d.get("configs") or {}).get(key) or {}
        if "error" in o or not o:
            out.append(f"| {name} | — | — | {o.get('error', 'not run')} | | | | | |")
            continue
        for leg, label in (("gpu_native_loop", "**GPU, the values in device memory, the library's own loop** (`gp_lm_graph_optimize`: linearise | damped step + retract + error evaluation, ONE wait per trial)"),
                           ("gpu_device_trial", "GPU, the values in device memory, the interpreter driving `gp_lm_graph_linearize` / `_try_lambda` / `_accept` (solve = the whole trial; linearise = its issue, + its wait in the split run)"),
                           ("gpu_device_solve", "GPU, host-driven (rounds 4 – 5's form): poses up, linearise | `gp_sparse_system_step`, wait | numpy retract, poses up, error evaluation, wait"), ("gpu_host_solve", "GPU linearise / error, numpy solve on the host"),
                           ("cpu_baseline", f"the reference's CPU factor ({o['cpu_baseline']['cores']} threads) + numpy solve: {o['cpu_baseline'].get('sample', '')[:60]}")):
            if leg not in o:
                continue
            x = o[leg]
            ph = x.get("ms_per_iteration_by_phase")
            if not ph:  # (the library's loop never returns to the interpreter between its phases)
                out.append(f"| {name} | {label} | {x['iterations']} ({x['inner_iterations']}) | **{x['ms_per_iteration']:.4f}** | — | — | — | — | "
                           f"{'met' if x['gate_met'] else 'NOT met'}: {x['max_rotation_error_rad']:.5f} rad / {x['max_translation_error_m']:.4f} m |")
                continue
            out.append(f"| {name} | {label} | {x['iterations']} ({x['inner_iterations']}) | **{x['ms_per_iteration']:.4f}** | {ph['linearize']:.4f} | {ph['solve']:.4f} | {ph['error']:.4f} | {ph['glue']:.4f} | "
                       f"{'met' if x['gate_met'] else 'NOT met'}: {x['max_rotation_error_rad']:.5f} rad / {x['max_translation_error_m']:.4f} m |")
    out.append("")
    out.append(f"Across the runs of the driver's command: C3 **{rng([x['legs'].get('lm_c3_ms_iter') for x in rs], '{:.3f}')} ms per iteration** in the library's loop ({rng([x['legs'].get('lm_c3_trial_ms_iter') for x in rs], '{:.3f}')} with the interpreter "
               f"driving the three calls, {rng([x['legs'].get('lm_c3_host_driven_ms_iter') for x in rs], '{:.3f}')} host-driven: solve {rng([x['legs'].get('lm_c3_solve_ms') for x in rs], '{:.3f}')}), C1 {rng([x['legs'].get('lm_c1_ms_iter') for x in rs], '{:.3f}')} "
               f"({rng([x['legs'].get('lm_c1_host_driven_ms_iter') for x in rs], '{:.3f}')} host-driven).  Round 5's driver run: C3 0.451 (solve 0.226).  VERDICT r05 #4 asked for ≤ 0.30.")
    return "\n".join(out)


def solver():
    rows = jl("r06_solver_step_time.jsonl")
    out = ["| graph (structure of the damped system) | ordering | levels / critical columns / blocks of L | assembly + ONE launch (`sparse_small_step_kernel`; the product: a team of waves per list, pipelined) ms | … a lone wave per list ms | … its first form (teams of waves in lock step) ms | multi-launch ms | bit-identical |",
           "|---|---|---|---|---|---|---|---|"]
    for x in rows:
        if x["ordering"] not in ("auto", "natural"):
            continue
        f = lambda k: f"{x[k]:.4f}" if x.get(k) is not None else "—"  # noqa: E731
        same = (x.get("bit_identical") and x.get("teams_bit_identical", True) and x.get("lone_waves_bit_identical", True)) if x.get("bit_identical") is not None else "—"
        out.append(f"| {x['graph']} | {x['ordering']} | {x['levels']} / {x['critical_columns']} / {x['l_blocks']} | " + (f"**{x['one_launch_ms']:.4f}**" if x.get("one_launch_ms") is not None else "does not fit the LDS") +
                   f" | {f('one_launch_lone_waves_ms')} | {f('one_launch_teams_ms')} | {x['multi_launch_ms']:.4f} | {same} |")
    tr = jl("r06_solver_trace.jsonl")
    if tr:
        out.append("")
        out.append("Thread 0's stamps on configs[2]'s structure (`scripts/r06/solver_trace.py`, shader clocks): " + "; ".join(
            f"{ {'True': 'the product', 'lone-waves': 'a lone wave per list', 'teams': 'teams in lock step'}.get(x['form'], x['form']) }: kernel {x['phase_clocks']['end'] / 1e3:.0f} k (factorisation {(x['phase_clocks']['factored'] - x['phase_clocks']['system_in_lds']) / 1e3:.0f} k, "
            f"backward substitution {(x['phase_clocks']['substituted'] - x['phase_clocks']['factored']) / 1e3:.0f} k), a steady column {x['first_level_rounds_clocks_gather_diag_below_next'][8][3]}" for x in tr) + ".")
    return "\n".join(out)


SECTIONS = {"headline": headline, "results": results, "lm": lm, "solver": solver}


def main():
    blocks = {k: f() for k, f in SECTIONS.items()}
    if "--write" not in sys.argv:
        for k, v in blocks.items():
            print(f"<!-- r06:{k}:begin -->\n{v}\n<!-- r06:{k}:end -->\n")
        return
    for doc in ("DESIGN.md", "README.md", "BASELINE.md"):
        path = os.path.join(ROOT, doc)
        text = open(path, encoding="utf-8").read()
        for k, v in blocks.items():
            text = re.sub(r"(<!-- r06:%s:begin -->\n).*?(\n<!-- r06:%s:end -->)" % (k, k), lambda m: m.group(1) + v + m.group(2), text, flags=re.S)
        open(path, "w", encoding="utf-8").write(text)
    print("written")


if __name__ == "__main__":
    main()
