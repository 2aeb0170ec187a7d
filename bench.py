#!/usr/bin/env python
"""bench.py -- VGICP linearise throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[1] -- ONE VGICP factor per GPU, 1 M synthetic source points vs a 2 M-point GaussianVoxelMap at 0.5 m
(gtsam_points_amd.synthetic.make_c2_workload; rank r uses seed 42 + r).  A "step" is one linearize() pass as the optimizer sees it.
  N = 1: gp_vgicp_batch_linearize -- pose in host memory -> ONE launch (the stream kernel's last workgroups finalize) -> record in host memory, synchronised.
  N > 1: pose -> stream kernel -> finalize kernel -> ONE exchange of the ranks' records -> records of ALL ranks in host memory, synchronised.  The exchange is
         --exchange all_reduce by default: ONE RCCL all-reduce (SUM) over the zeroed stacked [N x 122] f64 buffer, the collective BASELINE.json's north star names.
         In the SAME job the K steps are then repeated with the other two forms (in-place RCCL all-gather; direct peer stores over xGMI, csrc/gp_peer.hip) and
         reported as exchange_ms{all_reduce, all_gather, peer}; every form's exchanged stack is verified bit for bit against SHA-256 digests of what each rank
         computed (exchange_verified), and rank 0 holds its own row to the CPU oracle.
Inputs (source cloud, voxel map) are resident in HBM before the timed region.  value = N * 1e6 * K / elapsed.  Weak scaling.

Output: the LAST stdout line is ONE compact JSON object (< 4 KB, numbers and short names only: compact_result()).  Everything else -- the notes, BASELINE configs[0], [2],
[3], [4], the LM loops, the map build, the 8 M-point source -- goes to bench_detail.json next to this file (bench_detail.py holds those legs); the line carries one
number per leg under `legs`.  The optional legs run while --budget-seconds lasts (default 45 s of wall for the whole run; --detail = no limit).

  roofline     -- dominant kernel (vgicp_stream_kernel): algorithmic bytes (48 N_src + 16 N_buckets + 52 N_voxels + 560) / its mean duration.
                  frac / kernel_ms: the WHOLE fused kernel as the K timed steps ran it, by its own 100 MHz stamps; frac_streaming: its streaming slice;
                  frac_rocprof / rocprof_avg_ms: rocprofv3 --kernel-trace --stats over a child run of the same protocol, measured in THIS run (the figure a reader of
                  profiles/*kernel_stats.csv computes); traffic: HBM bytes per launch from in-run rocprofv3 --pmc passes.  Peak 8 TB/s HBM3E.
  cpu_baseline -- the reference's own CPU factor (oracle/_ref/libref.so, kind "reference"; the C restatement, kind "port", when that library is absent) timed on
                  this box's cores on the same workload, rank 0 / N = 1 only.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench_detail import HBM_PEAK_GBS, PhaseGuard, pick_cpu_threads  # noqa: E402

METRIC = "point-correspondences/sec (VGICP linearize, 1M-pt source vs 2M-pt voxel map)"
MAX_LINE_BYTES = 4096
PARITY_GATE = 1e-5  # BASELINE.json north_star: <= 1e-5 relative on H and b against the reference CPU factor
DETAIL_FILE = "bench_detail.json"
BLOCKS = ["H_target", "H_source", "H_target_source", "b_target", "b_source"]
KERNEL_NAMES = {12: "vgicp_stream_kernel", 8: "vgicp_pipeline_kernel<look-ahead>", 2: "vgicp_pipeline_kernel<hashed>"}
# what the last line must carry (tests/test_bench_line_cpu.py holds compact_result() to this)
REQUIRED_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline", "parity_max", "detail_file"]
REQUIRED_CONFIG = ["workload", "source_points", "target_points", "resolution", "num_voxels", "num_buckets", "exchange"]
REQUIRED_ROOFLINE = ["bound", "kernel", "achieved", "peak", "unit", "frac", "frac_streaming", "frac_rocprof", "kernel_ms", "algorithmic_bytes", "traffic", "rocprof_avg_ms"]
REQUIRED_CPU = ["value", "unit", "cores", "kind", "sample", "ms_per_linearize"]


def _clean(x):
    """strict JSON: NaN / Infinity become null, numpy scalars become Python numbers"""
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        x = x.item()
    if isinstance(x, float) and not math.isfinite(x):
        return None
    return x


def _pick(d, keys):
    d = d or {}
    return {k: d.get(k) for k in keys}


def compact_result(full):
    """The object of the last stdout line: the contract's keys, numbers and short names only -- no prose.  One number per optional leg under `legs`."""
    r, cfg, cpu = full.get("roofline") or {}, full.get("config") or {}, full.get("cpu_baseline")
    out = {k: full.get(k) for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]}
    out["config"] = _pick(cfg, REQUIRED_CONFIG + ["inlier_fraction", "device_warmup_ms", "device_warmup_steps"])
    out["roofline"] = _pick(r, REQUIRED_ROOFLINE + ["kernel_ms_back_to_back", "frac_back_to_back", "streaming_ms", "rocprof_calls"])
    out["cpu_baseline"] = _pick(cpu, REQUIRED_CPU + ["cores_available", "ms_per_linearize_1thread"]) if cpu else None
    par = full.get("parity_vs_oracle")
    vals = [par.get(k) for k in BLOCKS + ["error"]] if par else []
    finite = bool(vals) and all(isinstance(v, (int, float)) and math.isfinite(v) for v in vals)
    out["parity_max"] = max(vals) if finite else None  # (a non-finite entry is a failure, not a maximum: parity_ok says so)
    out["parity_inliers_equal"] = par.get("num_inliers_equal") if par else None
    out["parity_ok"] = bool(finite and max(vals) <= PARITY_GATE and par.get("num_inliers_equal")) if par else None
    for k in ["exchange_ms", "exchange_verified", "rccl_world", "backend"]:
        if full.get(k) is not None:
            out[k] = full[k]
    legs = {}

    def put(name, obj, *path):
        for p in path:
            obj = obj.get(p) if isinstance(obj, dict) else None
        if isinstance(obj, (int, float)) and not isinstance(obj, bool):
            legs[name] = obj

    c = full.get("configs") or {}
    put("C1_ms", c, "C1", "ms")
    put("C1_parity", {"x": max([v for k, v in ((c.get("C1") or {}).get("parity_vs_reference") or {}).items() if k != "num_inliers_equal"], default=None)}, "x")
    put("C3_ms", c, "C3", "ms")
    put("C3_parity", c, "C3", "parity_vs_reference_max")
    put("C3_cpu_ms", c, "C3", "cpu_baseline", "ms")
    put("C5_cov_ms", c, "C5", "covariances", "ms")
    put("C5_cov_frac", c, "C5", "covariances", "roofline", "frac")
    put("C5_gicp_ms", c, "C5", "gicp", "ms")
    put("map_build_ms", c, "map_build", "ms")
    put("map_build_frac", c, "map_build", "roofline", "frac")
    # lm_*_ms_iter: the library's own loop over the device-resident trial (gp_lm_graph_optimize); _host_driven: linearise / step / numpy retract / error evaluation as four host calls
    put("lm_c1_ms_iter", c, "lm_c1", "gpu_native_loop", "ms_per_iteration")
    put("lm_c3_ms_iter", c, "lm_c3", "gpu_native_loop", "ms_per_iteration")
    put("lm_c3_trial_ms_iter", c, "lm_c3", "gpu_device_trial", "ms_per_iteration")
    put("lm_c1_host_driven_ms_iter", c, "lm_c1", "gpu_device_solve", "ms_per_iteration")
    put("lm_c3_host_driven_ms_iter", c, "lm_c3", "gpu_device_solve", "ms_per_iteration")
    put("lm_c3_solve_ms", c, "lm_c3", "gpu_device_solve", "ms_per_iteration_by_phase", "solve")
    put("lm_c3_cpu_ms_iter", c, "lm_c3", "cpu_baseline", "ms_per_iteration")
    put("big_source_frac", full, "big_source", "roofline", "frac")
    put("big_source_ms", full, "big_source", "ms_per_linearize")
    put("c4_ms", full, "c4", "ms_per_linearize")
    put("c4_value", full, "c4", "value")
    if isinstance((full.get("c4") or {}).get("exchange_verified"), bool):
        legs["c4_verified"] = full["c4"]["exchange_verified"]
    out["legs"] = legs
    out["legs_skipped"] = full.get("legs_skipped") or []
    out["run_seconds"] = full.get("run_seconds")
    out["detail_file"] = full.get("detail_file", DETAIL_FILE)

    def short(x):  # (names, not notes: whatever a leg wrote into a string field, the line keeps its first 96 characters)
        if isinstance(x, dict):
            return {k: short(v) for k, v in x.items()}
        if isinstance(x, list):
            return [short(v) for v in x]
        return x[:96] if isinstance(x, str) else x

    return short(_clean(out))


def compact_line(full):
    """json text of compact_result(full): strict JSON, one line, shorter than MAX_LINE_BYTES (raises otherwise -- a line the driver cannot parse is no measurement)"""
    line = json.dumps(compact_result(full), allow_nan=False, separators=(",", ":"))
    if len(line.encode()) >= MAX_LINE_BYTES or "\n" in line:
        raise RuntimeError(f"bench.py: the result line is {len(line.encode())} bytes (limit {MAX_LINE_BYTES})")
    return line


def _effective_kernel(lib, batch):
    v = C.c_int(-1)
    lib.gp_vgicp_batch_get_tuning(batch, 6, C.byref(v))  # GP_TUNE_EFFECTIVE_KERNEL
    return v.value


def _rocprof_child(extra, child_args, timeout):
    """runs `rocprofv3 <extra> -- python bench.py <child_args>` in /tmp; returns (output directory object, process) -- the caller parses the csv files"""
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, None
    tmp = tempfile.TemporaryDirectory(prefix="gp_prof_", dir="/tmp")
    cmd = [exe] + extra + ["--output-format", "csv", "-d", tmp.name, "-o", "p", "--", sys.executable, os.path.abspath(__file__)] + child_args
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    return tmp, p


def _csv_rows(tmpdir, suffix):
    import csv

    for dirpath, _dirs, files in os.walk(tmpdir):
        for fn in files:
            if fn.endswith(suffix):
                with open(os.path.join(dirpath, fn)) as f:
                    yield from csv.DictReader(f)


def _workload_args(args):
    return ["--source-points", str(args.source_points), "--target-points", str(args.target_points), "--resolution", str(args.resolution)]


def measure_traffic(args):
    """roofline.traffic measured IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only, as MI355X_MICROARCH.md's HBM
    section prescribes) over a small child run of this file (--pmc-child: 7 fused steps of the same workload + a calibration stream of known 48 N bytes in the stream
    kernel's own access pattern), read side scaled on the calibration stream (rocprofv3's FETCH_SIZE prices a 128-B request at 64 B on gfx950: DESIGN.md 6).
    Never raises; returns {"error": ...} when rocprofv3 is absent or a pass fails."""
    from collections import defaultdict

    means = {}
    try:
        t0 = time.time()
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            tmp, p = _rocprof_child(["--pmc", ctr, "--kernel-trace"], ["--pmc-child"] + _workload_args(args), 150)
            if tmp is None:
                return dict(error="rocprofv3 not on PATH")
            with tmp:
                acc = defaultdict(list)
                for row in _csv_rows(tmp.name, "counter_collection.csv"):
                    if row.get("Counter_Name") == ctr:
                        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
            if not acc:
                return dict(error=f"rocprofv3 --pmc {ctr}: no counter rows (exit code {p.returncode}): {p.stderr[-300:]}")
            means[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

        def find(d, needle):
            return next((v for k, v in d.items() if needle in k), None)

        calib, tile_f, tile_w = find(means["FETCH_SIZE"], "calibration_stream_kernel"), find(means["FETCH_SIZE"], "vgicp_stream_kernel"), find(means["WRITE_SIZE"], "vgicp_stream_kernel")
        if not calib or not tile_f:
            return dict(error="the profiled child ran no calibration / stream kernel")
        scale = 48.0 * args.source_points / (calib[0] * 1024.0)
        return dict(hbm_bytes_per_launch=int(tile_f[0] * 1024.0 * scale + (tile_w[0] if tile_w else 0.0) * 1024.0), fetch_size_kib=round(tile_f[0], 1),
                    write_size_kib=round(tile_w[0], 1) if tile_w else None, launches=tile_f[1], calibration_fetch_kib=round(calib[0], 1), fetch_scale=round(scale, 4),
                    seconds=round(time.time() - t0, 1),
                    source="measured in THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, --kernel-trace only) over a child run of 7 fused steps of the same "
                           "workload; read side scaled on a calibration stream of known bytes (fetch_scale); per launch of vgicp_stream_kernel")
    except Exception as exc:
        return dict(error=f"{type(exc).__name__}: {exc}")


def measure_rocprof(args):
    """roofline.rocprof_avg_ms measured IN THIS RUN: rocprofv3 --kernel-trace --stats over a child run of this file that runs the headline's own protocol (device wake-up, W
    warm-up steps, K timed steps: --profile-child) and nothing else; the AverageNs of vgicp_stream_kernel in its kernel_stats.csv -- the number a reader of the committed
    profiles/*kernel_stats.csv computes.  Never raises."""
    try:
        t0 = time.time()
        tmp, p = _rocprof_child(["--kernel-trace", "--stats"], ["--profile-child", "--steps", str(args.steps), "--warmup", str(args.warmup), "--device-warmup-ms", str(args.device_warmup_ms)]
                                + _workload_args(args), 150)
        if tmp is None:
            return dict(error="rocprofv3 not on PATH")
        with tmp:
            row = next((r for r in _csv_rows(tmp.name, "kernel_stats.csv") if "vgicp_stream_kernel" in r.get("Name", "")), None)
        if row is None:
            return dict(error=f"no vgicp_stream_kernel row in kernel_stats.csv (exit code {p.returncode}): {p.stderr[-300:]}")
        return dict(avg_ms=round(float(row["AverageNs"]) * 1e-6, 6), calls=int(row["Calls"]), min_ms=round(float(row["MinNs"]) * 1e-6, 6), max_ms=round(float(row["MaxNs"]) * 1e-6, 6),
                    seconds=round(time.time() - t0, 1), source="measured in THIS run: rocprofv3 --kernel-trace --stats over a child run of the headline protocol (wake-up + W + K steps)")
    except Exception as exc:
        return dict(error=f"{type(exc).__name__}: {exc}")


def main():
    t_run = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--source-points", type=int, default=1_000_000)
    ap.add_argument("--target-points", type=int, default=2_000_000)
    ap.add_argument("--resolution", type=float, default=0.5)
    ap.add_argument("--budget-seconds", type=float, default=45.0, help="wall-clock budget of the whole run: an optional leg starts only while its estimate still fits")
    ap.add_argument("--detail", action="store_true", help="run every optional leg whatever it costs (no budget; the LM legs with the full CPU loop and all three solvers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="CPU baseline: seconds of repeated linearize() passes (bounded sample)")
    ap.add_argument("--kernel-iters", type=int, default=50)
    ap.add_argument("--no-c4", action="store_true", help="skip the sharded 4096-factor configuration (BASELINE configs[3])")
    ap.add_argument("--c4-steps", type=int, default=30)
    ap.add_argument("--c4-exchange", choices=["all_gather", "all_reduce"], default="all_reduce", help="the c4 step's collective (all_gather: in place, falls back to the all-reduce when the shards are unequal)")
    ap.add_argument("--exchange", choices=["all_reduce", "all_gather", "peer"], default="all_reduce",
                    help="N > 1: the exchange the HEADLINE steps use.  all_reduce (default) = ONE RCCL all-reduce over the zeroed [N x 122] stack, the north star's collective; "
                         "all_gather = RCCL, in place; peer = direct stores into every peer's buffer over xGMI (csrc/gp_peer.hip).  The other forms are timed in the same job (exchange_ms)")
    ap.add_argument("--no-exchange-forms", action="store_true", help="N > 1: time only the headline's exchange form, not all three")
    ap.add_argument("--no-c4-inlib", action="store_true", help="skip the single-process multi-device leg of c4")
    ap.add_argument("--c4-inlib-only", action="store_true", help="(internal) run only the single-process multi-device leg of c4 and print its JSON object")
    ap.add_argument("--finalize", choices=["fused", "two-kernel"], default="fused", help="N = 1 step: fused = the library default (one launch); two-kernel = GP_TUNE_FUSED_FINALIZE 0")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1 / C3 / C5 / map-build / LM legs (BASELINE configs[0], [2], [4])")
    ap.add_argument("--device-warmup-ms", type=float, default=300.0, help="untimed: run the step for this long before the W warm-up steps (0 = off); ms_per_step_cold is the figure without it")
    ap.add_argument("--no-big-source", action="store_true", help="skip the 8 M-point source (beyond the Infinity Cache) leg")
    ap.add_argument("--no-mirror", action="store_true", help="A/B: stream the caller's 12 + 36 B per point instead of the packed 36-B mirror (GP_TUNE_SOURCE_MIRROR 0)")
    ap.add_argument("--phase-seconds", type=float, default=120.0, help="N > 1: time box of every distributed phase: a hung rank ends the job")
    ap.add_argument("--no-cold", action="store_true", help="skip the ms_per_step_cold leg (the K timed steps without the device wake-up in front)")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic in this run (rocprofv3 --pmc passes of a small child run)")
    ap.add_argument("--no-rocprof", action="store_true", help="do not measure roofline.rocprof_avg_ms in this run (rocprofv3 --kernel-trace --stats over a child run)")
    ap.add_argument("--no-lm", action="store_true", help="skip the lm_c1 / lm_c3 legs")
    ap.add_argument("--lm-full-cpu", action="store_true", help="LM legs: the whole CPU loop and all three solver forms (default: a 3-iteration CPU sample, two forms)")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) the small run the --pmc passes profile: 7 fused steps + the calibration stream, no output line")
    ap.add_argument("--profile-child", action="store_true", help="(internal) the headline protocol only (wake-up, W warm-up, K steps), no output line: what measure_rocprof profiles")
    ap.add_argument("--inlib-devices", type=int, default=0, help="(internal) devices the --c4-inlib-only leg drives (0 = all visible)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, DETAIL_FILE))
    args = ap.parse_args()
    if args.detail:
        args.budget_seconds, args.lm_full_cpu = 1e9, True

    import torch
    import torch.distributed as dist

    import bench_detail

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if args.c4_inlib_only:
        import gtsam_points_amd as gpa
        from gtsam_points_amd import _capi, synthetic

        torch.cuda.set_device(0)
        lib = gpa.load()
        _capi.check(lib.gp_set_device(0), "gp_set_device")
        print(json.dumps(_clean(bench_detail.run_c4_inlib(lib, gpa, _capi, synthetic, torch, torch.device("cuda:0"), max(args.c4_steps // 3, 5), args.inlib_devices))), flush=True)
        return
    dev_index = local_rank % torch.cuda.device_count()  # == local_rank on a full node; lets a 1-GPU box rehearse N > 1
    torch.cuda.set_device(dev_index)
    device = torch.device(f"cuda:{dev_index}")
    # GP_BENCH_FORCE_DIST=1: take the N > 1 code path with ONE rank -- how a 1-GPU box runs the RCCL initialisation and a 1-rank collective (tests/test_multi_gpu.py)
    dist_on = world > 1 or bool(os.environ.get("GP_BENCH_FORCE_DIST"))
    backend = None
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime

        backend = os.environ.get("GP_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; "gloo" only for the 1-GPU rehearsal
        box = datetime.timedelta(seconds=args.phase_seconds)
        with PhaseGuard(args.phase_seconds, "process group rendezvous"):
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=box)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=box)

    import gtsam_points_amd as gpa
    from gtsam_points_amd import _capi, synthetic

    lib = gpa.load()
    _capi.check(lib.gp_set_device(dev_index), "gp_set_device")

    # ---- workload, resident in HBM before timing ----
    t_gen = time.time()
    d = synthetic.make_c2_workload(args.source_points, args.target_points, seed=42 + rank)
    t_gen = time.time() - t_gen
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"], device=device)
    src = gpa.PointCloudGPU(d["source_points"], d["source_covs"], device=device)
    torch.cuda.synchronize()
    t_map = time.time()
    vm = gpa.GaussianVoxelMapGPU(args.resolution, target_points_drop_rate=0.0)
    vm.insert(tgt)
    t_map = time.time() - t_map
    info = vm.voxelmap_info
    stream = torch.cuda.current_stream(device)
    sptr = C.c_void_p(stream.cuda_stream)
    factor = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src, stream=sptr)
    arr = (C.c_void_p * 1)(factor._h.value)
    batch = C.c_void_p()
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, sptr, C.byref(batch)), "gp_vgicp_batch_create")
    if args.finalize == "two-kernel":
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_FUSED_FINALIZE, 0), "finalize form")
    if args.no_mirror:
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_SOURCE_MIRROR, 0), "source mirror")
    delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()

    REC = _capi.LINEARIZED6_DOUBLES
    host_out = torch.zeros((world, REC), dtype=torch.float64).pin_memory()
    out_np = host_out.numpy()
    sharded_by_form, sharded = {}, None

    if not dist_on:
        # the product's synchronous entry point: pose in host memory -> records in host memory
        linearize = lib.gp_vgicp_batch_linearize  # bound once: the step is ~20 us, attribute lookups and .ctypes views are not free
        pose_ptr, out_ptr = C.c_void_p(pose.ctypes.data), C.c_void_p(out_np.ctypes.data)

        def step():
            if linearize(batch, pose_ptr, out_ptr) != 0:
                _capi.check(1, "gp_vgicp_batch_linearize")

    else:
        from gtsam_points_amd.distributed import ShardedLinearizer

        issue_linearize = lib.gp_vgicp_batch_issue_linearize
        pose_ptr = C.c_void_p(pose.ctypes.data)
        row_ptr = {}

        def issue(poses_local, view):  # (the row views are few objects: their addresses are taken once)
            p = row_ptr.get(id(view))
            if p is None:
                p = row_ptr[id(view)] = (C.c_void_p(view.data_ptr()), view)  # (the view is kept: its id stays its own)
            if issue_linearize(batch, pose_ptr, p[0]) != 0:
                _capi.check(1, "gp_vgicp_batch_issue_linearize")

        def make_sharded(form):
            s = ShardedLinearizer(world, (rank, rank + 1), device, issue, always_exchange=True, exchange=form, host_out=host_out)
            with PhaseGuard(args.phase_seconds, f"exchange set-up ({form})"):  # (the first pass decides the exchange, collectively)
                s.linearize(pose)
                torch.cuda.synchronize()
                s.check()
            return s

        sharded = sharded_by_form[args.exchange] = make_sharded(args.exchange)

        def make_step(s):
            delivers = s.delivers_to_host

            def step_():
                # local kernels into the rank's row, then ONE exchange: one RCCL collective (zero + all-reduce; or the in-place all-gather) and a D2H copy, or direct
                # stores into the peers' buffers over xGMI (the exchange kernel also fills host_out)
                stacked = s.linearize(pose)
                if not delivers:
                    host_out.copy_(stacked, non_blocking=True)
                stream.synchronize()

            return step_

        step = make_step(sharded)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:  # the run the --pmc passes profile (measure_traffic): a few fused steps + the calibration stream of known bytes, nothing else
        for _ in range(7):
            step()
        barrier()
        _capi.check(_capi.load_tune().gp_debug_calibration_stream(src.ptr(src.points_gpu), src.ptr(src.covs_gpu), args.source_points, 5, sptr), "calibration")
        torch.cuda.synchronize()
        lib.gp_vgicp_batch_destroy(batch)
        return None

    def timed_steps(label, step_fn=None, s=None):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks"""
        step_fn = step_fn or step
        s = s or sharded
        with PhaseGuard(args.phase_seconds if dist_on else 600.0, label):
            for _ in range(args.warmup):
                step_fn()
            barrier()
            lib.gp_vgicp_batch_device_times(batch, 1, None, None, None)  # reset: the kernel's own time stamps of the timed steps only
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_fn()
            barrier()
            el = time.perf_counter() - t0
            if dist_on:
                s.check()  # (peer exchange: a peer that did not arrive within the kernel's time box is an error, not a stale stack)
            n_, su_, ku_ = C.c_double(), C.c_double(), C.c_double()
            lib.gp_vgicp_batch_device_times(batch, 0, C.byref(n_), C.byref(su_), C.byref(ku_))
            if dist_on:
                te = torch.tensor([el], dtype=torch.float64, device=device)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                el = float(te.item())
        return el, n_, su_, ku_

    def wake_up():
        """device wake-up (untimed, before the W warm-up steps): seconds of host-side set-up leave the device in a low power state and it takes ~10 ms of work before the step
        settles (profiles/r04_warm.jsonl); the same synchronous step is run for --device-warmup-ms first.  Returns the number of steps run."""
        t_wake, n = time.perf_counter(), 0
        with PhaseGuard(args.phase_seconds if dist_on else 600.0, "device wake-up"):
            if dist_on:
                # a step holds a collective: every rank must run the SAME number of them.  Chunks of 50 steps; after each the ranks agree (MAX over ranks of the time spent)
                spent = torch.zeros(1, dtype=torch.float64, device=device)
                while n < 5000 and args.device_warmup_ms > 0:
                    for _ in range(50):
                        step()
                        n += 1
                        if n % 25 == 0:
                            torch.cuda.synchronize()
                    spent[0] = (time.perf_counter() - t_wake) * 1e3
                    dist.all_reduce(spent, op=dist.ReduceOp.MAX)
                    if float(spent.item()) >= args.device_warmup_ms:
                        break
            else:
                while (time.perf_counter() - t_wake) * 1e3 < args.device_warmup_ms:
                    step()
                    n += 1
                    if n % 25 == 0:
                        torch.cuda.synchronize()  # (the timed region is bracketed by device synchronisations: the wake-up runs the same pattern)
        return n

    if args.profile_child:  # the headline protocol and nothing else (measure_rocprof)
        wake_up()
        timed_steps("timed steps (profiled child)")
        lib.gp_vgicp_batch_destroy(batch)
        return None

    # ms_per_step_cold: the same W + K protocol with NOTHING in front -- the device as seconds of host-side set-up left it (rounds 1-3's protocol)
    cold = None
    if not args.no_cold:
        el_c, n_c, su_c, ku_c = timed_steps("timed steps (cold)")
        cold = dict(ms_per_step=round(el_c / args.steps * 1e3, 5), stream_us=round(su_c.value, 3) if n_c.value >= args.steps else None,
                    fused_kernel_us=round(ku_c.value, 3) if n_c.value >= args.steps else None)
    wake_steps = wake_up()
    elapsed, dev_steps, dev_stream_us, dev_kernel_us = timed_steps("timed steps")
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.source_points * args.steps / elapsed

    # ---- N > 1: the other exchange forms in the same job, and the bit-for-bit check of every form's exchanged stack ----
    exchange_ms, exchange_verified, verify_detail, exchange_notes = None, None, None, None
    if dist_on and rank == 0:
        # from here on the phases are optional and, across devices, run for the first time on the driver's node: if one of them hangs and the job is torn down, the headline
        # measured above still leaves as the job's last stdout line (bench_detail.register_last_resort; not printed by a job that ends normally)
        bench_detail.register_last_resort(compact_line(dict(
            metric=METRIC, value=round(value, 1), unit="point-correspondences/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 5),
            ms_per_step_cold=cold["ms_per_step"] if cold else None, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
            config=dict(workload="BASELINE configs[1]: single VGICP factor per GPU, 1M synthetic source pts vs 2M-pt GaussianVoxelMap @0.5 m", source_points=args.source_points, target_points=args.target_points,
                        resolution=args.resolution, num_voxels=info.num_voxels, num_buckets=info.num_buckets, exchange=sharded.exchange, device_warmup_ms=args.device_warmup_ms, device_warmup_steps=wake_steps),
            roofline=dict(bound="hbm", kernel="vgicp_stream_kernel", peak=HBM_PEAK_GBS, unit="GB/s"), cpu_baseline=None, parity_vs_oracle=None, backend=dist.get_backend(), rccl_world=dist.get_world_size(),
            exchange_ms={args.exchange: round(ms_per_step, 5)}, legs_skipped=["everything behind the headline"], run_seconds=round(time.time() - t_run, 1))))
    if dist_on:
        from gtsam_points_amd.distributed import verify_exchanged_stack

        exchange_ms, exchange_notes = {args.exchange: round(ms_per_step, 5)}, {}
        forms = [args.exchange] if args.no_exchange_forms else [args.exchange] + [f for f in ("all_reduce", "all_gather", "peer") if f != args.exchange]
        def all_agree(ok):  # (one tiny collective: every rank takes the same branch behind a failure)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        for form in forms[1:]:
            # A form that fails -- buffers that cannot be shared, a peer that does not arrive within the exchange kernel's time box (which poisons every rank's next wait: all
            # ranks fail in the same place, behind the steps' barrier) -- must not take the headline's line with it: the ranks agree, the form reads null, the reason goes to the
            # detail file.  (A rank that HANGS is the phase guard's business.)
            s_f, why = None, None
            try:
                s_f = make_sharded(form)
                if s_f.exchange != form:  # (a form every rank fell back from together)
                    why = f"ran as {s_f.exchange}: {s_f.peer_note or 'the plan or the backend does not take this form'}"
                else:
                    el_f, _, _, _ = timed_steps(f"timed steps ({form})", make_step(s_f), s_f)
            except Exception as exc:
                why = f"{type(exc).__name__}: {exc}"
            if all_agree(why is None):
                sharded_by_form[form] = s_f
                exchange_ms[form] = round(el_f / args.steps * 1e3, 5)
            else:
                exchange_ms[form] = None
                exchange_notes[form] = why or "failed on another rank"
                if s_f is not None:
                    try:
                        s_f.close()
                    except Exception:
                        pass
        # what this rank computed, by itself: the batch's kernels into a private buffer, no exchange (the kernels are bit-reproducible: tests/test_vgicp_gpu.py)
        own = torch.zeros((1, REC), dtype=torch.float64, device=device)
        issue(pose, own)
        torch.cuda.synchronize()
        own_host = own.cpu().numpy()
        verify_detail, exchange_verified = {}, True
        with PhaseGuard(args.phase_seconds, "exchange verification"):
            for form, s in sharded_by_form.items():
                if exchange_ms.get(form) is None:
                    continue
                host_out.zero_()
                err = None
                try:
                    make_step(s)()
                    s.check()
                except Exception as exc:  # (the stack stays zero: the verification below fails for this form on every rank, which is the right verdict)
                    err = f"{type(exc).__name__}: {exc}"
                ok, bad = verify_exchanged_stack(out_np, own_host, rank, rank + 1)
                verify_detail[form] = dict(verified=ok, bad_rows_by_rank=bad, error=err)
                exchange_verified = exchange_verified and ok
        step()  # (host_out holds the headline form's stack again)

    # ---- dominant-kernel roofline ----
    # (1) as the timed steps ran it: the fused kernel stamps its own start / last row in / sums out on the device's 100 MHz clock (gp_vgicp_batch_device_times)
    # (2) back to back: HIP events on the launch stream over a loop of stream-kernel launches (two-kernel form: the streaming part alone)
    ms_total, ms_main, ms_fin = C.c_float(), C.c_float(), C.c_float()
    _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, args.kernel_iters, C.byref(ms_total), C.byref(ms_main), C.byref(ms_fin)), "time_linearize")
    alg_bytes = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
    actual_bytes = int(lib.gp_vgicp_batch_actual_bytes(batch))
    mirrored = C.c_int(-1)
    lib.gp_vgicp_batch_get_tuning(batch, _capi.GP_TUNE_EFFECTIVE_MIRROR, C.byref(mirrored))
    in_step = dev_steps.value >= args.steps and dev_stream_us.value > 0 and dev_kernel_us.value > 0
    streaming_ms = dev_stream_us.value * 1e-3 if in_step else ms_main.value
    kernel_ms = dev_kernel_us.value * 1e-3 if in_step else ms_main.value

    def _frac(ms):
        return round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms else None

    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    roofline = dict(
        bound="hbm", kernel=KERNEL_NAMES.get(_effective_kernel(lib, batch), "?"), achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5),
        frac_note=("frac = algorithmic bytes / the WHOLE fused kernel as the timed steps ran it (first workgroup started .. last part's sums handed to the host; the kernel's own "
                   "100 MHz stamps, mean over the K steps); frac_streaming = its streaming slice; frac_rocprof = algorithmic bytes / rocprofv3's average duration of the kernel")
        if in_step else "no fused steps in this configuration: frac is the stream kernel back to back under HIP events",
        frac_streaming=_frac(streaming_ms) if in_step else None, streaming_ms=round(streaming_ms, 5) if in_step else None,
        frac_rocprof=None, rocprof_avg_ms=None, rocprof_calls=None, traffic=None, traffic_source=None,
        algorithmic_bytes=alg_bytes,
        algorithmic_bytes_note="SURVEY.md 8(d), reference-layout accounting (48 B per source point + the reference's bucket table and voxel arrays): internal repacking does not change it",
        source_stream=("packed private mirror: 36 B per point (12 B point + the 6 floats of the symmetric covariance)" if mirrored.value == 1 else "the caller's arrays: 12 + 36 B per point"),
        actual_bytes=actual_bytes, frac_actual=round(actual_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        kernel_ms=round(kernel_ms, 5),
        kernel_ms_source=(f"measured in THIS run inside the {args.steps} timed steps: the whole fused kernel on the device's 100 MHz constant clock, stamped by the kernel itself "
                          "(gp_vgicp_batch_device_times); mean over the steps") if in_step else "measured in THIS run: HIP events over back-to-back stream-kernel launches on the launch stream",
        kernel_ms_back_to_back=round(ms_main.value, 5), frac_back_to_back=_frac(ms_main.value),
        step_finalize=args.finalize if not dist_on else "device-resident records (two kernels)", finalize_kernel_ms=round(ms_fin.value, 5), device_pass_ms=round(ms_total.value, 5), cold=cold)

    rec = gpa.LinearizedSystem6.from_doubles(host_out[rank].numpy().copy())

    # ---- the budget of the optional legs: the decision is rank 0's, taken for all ranks ----
    skipped = []

    def want(name, estimate):
        ok = (time.time() - t_run) + estimate <= args.budget_seconds
        if dist_on and world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
            dist.broadcast(flag, src=0)
            ok = bool(flag.item())
        if not ok:
            skipped.append(name)
        return ok

    # ---- rank 0: parity of its own row against the checker, and (N = 1) the CPU baseline ----
    cpu_baseline, parity = None, None
    if rank == 0 and not args.no_cpu_baseline:
        import oracle  # checker / baseline only -- never on the product path
        from oracle import refcapi

        avail = oracle.max_threads()
        use_ref = refcapi.available()  # the reference's own CPU sources (oracle/_ref/libref.so) when they were built
        VM_, VG_ = (refcapi.RefVoxelMap, refcapi.RefVGICPFactor) if use_ref else (oracle.OracleVoxelMap, oracle.OracleVGICPFactor)
        om = VM_(args.resolution)
        om.insert(d["target_points"], d["target_covs"])
        cores = pick_cpu_threads(avail, lambda c: VG_(om, d["source_points"], d["source_covs"], c), lambda o: o.linearize(delta), reps=2) if world == 1 else min(avail, 16)
        fo = VG_(om, d["source_points"], d["source_covs"], cores)
        Lo = fo.linearize(delta)  # warm-up + parity reference
        parity = {k: float(np.linalg.norm(getattr(rec, k) - getattr(Lo, k)) / np.linalg.norm(getattr(Lo, k))) for k in BLOCKS}
        parity["error"] = abs(rec.error - Lo.error) / abs(Lo.error)
        parity["num_inliers_equal"] = bool(rec.num_inliers == Lo.num_inliers)
        if world == 1 and not dist_on:
            times, t_start = [], time.perf_counter()
            while time.perf_counter() - t_start < args.cpu_seconds * 0.75 or len(times) < 3:
                t = time.perf_counter()
                fo.linearize(delta)
                times.append(time.perf_counter() - t)
            f1 = VG_(om, d["source_points"], d["source_covs"], 1)
            t1, t_start = [], time.perf_counter()
            while time.perf_counter() - t_start < args.cpu_seconds * 0.25 or len(t1) < 2:
                t = time.perf_counter()
                f1.linearize(delta)
                t1.append(time.perf_counter() - t)
            med = float(np.median(times))
            cpu_baseline = dict(
                value=round(args.source_points / med, 1), unit="point-correspondences/s", cores=cores, cores_available=avail, kind="reference" if use_ref else "port",
                sample=f"{len(times)} linearize() passes, {cores} threads", ms_per_linearize=round(med * 1e3, 3), ms_per_linearize_1thread=round(float(np.median(t1)) * 1e3, 3),
                cores_note="threads chosen by a probe over {all, 1/2, 32, 16} of the threads the box reports (pick_cpu_threads): the count that serves the reference's code best",
                sample_note=f"{len(times)} full linearize() passes of the same 1M-pt factor (median {med * 1e3:.2f} ms); 1 thread: {np.median(t1) * 1e3:.2f} ms over {len(t1)} passes")
        del fo, om

    # ---- optional legs, most valuable first, while the budget lasts ----
    single = rank == 0 and world == 1 and not dist_on
    traffic_detail = rocprof_detail = None
    if single and not args.no_rocprof and want("rocprof", 5.0):
        rocprof_detail = measure_rocprof(args)
        if rocprof_detail.get("avg_ms"):
            roofline.update(rocprof_avg_ms=rocprof_detail["avg_ms"], rocprof_calls=rocprof_detail["calls"], frac_rocprof=_frac(rocprof_detail["avg_ms"]))
    if single and not args.no_traffic and want("traffic", 8.5):
        traffic_detail = measure_traffic(args)
        if traffic_detail.get("hbm_bytes_per_launch"):
            roofline.update(traffic=traffic_detail["hbm_bytes_per_launch"], traffic_source=traffic_detail["source"])
    roofline.update(traffic_detail=traffic_detail, rocprof_detail=rocprof_detail)

    configs = big_source = c4 = None
    if single and not args.no_configs:
        try:
            configs = bench_detail.run_configs(args, lib, gpa, _capi, synthetic, torch, device, stream, want=want, target_cloud=tgt if args.target_points == 2_000_000 else None)
        except Exception as exc:  # the headline must survive an optional leg
            configs = dict(error=f"{type(exc).__name__}: {exc}")
    if single and not args.no_configs and not args.no_big_source and want("big_source", 6.0):
        try:
            big_source = bench_detail.run_big_source(args, lib, gpa, _capi, synthetic, torch, device, stream)
        except Exception as exc:  # (memory on a shared box)
            big_source = dict(error=f"{type(exc).__name__}: {exc}")
    if not args.no_c4 and want("c4", 14.0):
        c4 = bench_detail.run_c4(args, lib, gpa, _capi, synthetic, torch, dist, rank, world, device, stream, dist_on)

    result = None
    if rank == 0:
        if dist_on:
            exch_text = f"torch.distributed backend {dist.get_backend()}, world {dist.get_world_size()}, {sharded.exchange}"
            step_text = {"all_reduce": "pose -> stream kernel -> finalize kernel -> zero + ONE RCCL all-reduce (SUM) of the stacked [N x 122] f64 records -> D2H -> sync",
                         "all_gather": "pose -> stream kernel -> finalize kernel -> ONE in-place RCCL all-gather of the stacked [N x 122] f64 records -> D2H -> sync",
                         "peer": "pose -> stream kernel -> finalize kernel -> ONE exchange kernel: direct stores of the record into every peer's buffer over xGMI + the stack to pinned host memory -> sync"}.get(sharded.exchange, sharded.exchange)
        else:
            exch_text, step_text = None, "pose (host) -> ONE launch: stream kernel whose last workgroups finalize -> record in host memory, host polls the completion word"
        result = dict(
            metric=METRIC, value=round(value, 1), unit="point-correspondences/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 5),
            ms_per_step_cold=cold["ms_per_step"] if cold else None, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
            dtype_note="transform, fused covariance, its inverse, residual and all reductions in f64; the outer products after the inverse in f32 (kernel family GP_KERNEL_STREAM)",
            config=dict(workload="BASELINE configs[1]: single VGICP factor per GPU, 1M synthetic source pts vs 2M-pt GaussianVoxelMap @0.5 m", source_points=args.source_points,
                        target_points=args.target_points, resolution=args.resolution, num_voxels=info.num_voxels, num_buckets=info.num_buckets,
                        inlier_fraction=round(rec.num_inliers / args.source_points, 4), exchange=sharded.exchange if dist_on else None, exchange_detail=exch_text, step=step_text,
                        parallelism=f"{world} x 1 factor/GPU" if dist_on else "1 GPU", device_warmup_ms=args.device_warmup_ms, device_warmup_steps=wake_steps,
                        device_warmup_note="untimed, before the W warm-up steps: the same step run back to back until the device's power state has settled; ms_per_step_cold is the figure without it"),
            roofline=roofline, cpu_baseline=cpu_baseline, parity_vs_oracle=parity,
            backend=(dist.get_backend() if dist_on else None), rccl_world=(dist.get_world_size() if dist_on else None),
            exchange_ms=exchange_ms, exchange_verified=exchange_verified, exchange_verify_detail=verify_detail, exchange_notes=exchange_notes,
            c4=c4, configs=configs, big_source=big_source, legs_skipped=skipped, budget_seconds=args.budget_seconds,
            setup=dict(generate_s=round(t_gen, 2), voxelmap_build_s=round(t_map, 4)), detail_file=os.path.basename(args.detail_file))
    lib.gp_vgicp_batch_destroy(batch)
    if dist_on:
        with PhaseGuard(args.phase_seconds, "process group teardown"):
            for s in sharded_by_form.values():
                s.close()  # (peer exchange: the peers' buffers are unmapped behind a barrier)
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        if world > 1 and result.get("c4") is not None and not args.no_c4_inlib and (time.time() - t_run) + 30.0 <= args.budget_seconds:
            # the in-library path a C++ optimizer process uses (ONE process drives N devices) beside the torch.distributed step, per N.  Run when the ranks are gone, in a
            # process of its own with a time limit: neither a hang nor a crash of it may take the line with it
            import subprocess

            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--c4-inlib-only", "--inlib-devices", str(world), "--c4-steps", str(args.c4_steps)],
                                   capture_output=True, text=True, timeout=240)
                lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
                result["c4"]["inlib"] = json.loads(lines[-1]) if lines else dict(error=f"no result (exit code {p.returncode}): {p.stderr[-400:]}")
            except subprocess.TimeoutExpired:
                result["c4"]["inlib"] = dict(error="the in-library multi-device leg did not finish within 240 s and was stopped")
            except Exception as exc:
                result["c4"]["inlib"] = dict(error=f"{type(exc).__name__}: {exc}")
        result["run_seconds"] = round(time.time() - t_run, 1)
        try:
            with open(args.detail_file, "w") as f:
                json.dump(_clean(result), f, indent=1, allow_nan=False)
        except OSError as exc:  # (a read-only tree: the line still goes out)
            sys.stderr.write(f"bench.py: could not write {args.detail_file}: {exc}\n")
        # the record is the LAST stdout line: whatever a C library still holds in its stdio buffer (RCCL prints a version banner through printf, which would otherwise be
        # flushed at process exit, BEHIND this line) goes out first
        try:
            sys.stdout.flush()
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        bench_detail.register_last_resort(None)  # (the job ended normally: the real line follows)
        print(compact_line(result), flush=True)
    return result


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:  # (an exception behind the measured headline of an N > 1 job: the registered line still leaves -- bench_detail.register_last_resort)
        if not isinstance(exc, SystemExit) or exc.code not in (0, None):
            import bench_detail as _bd

            _bd.emit_last_resort(f"{type(exc).__name__}: {exc}")
        raise
