#!/usr/bin/env python
"""bench.py -- VGICP linearise throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[1] -- ONE VGICP factor per GPU, 1 M synthetic source points vs a
2 M-point GaussianVoxelMap at 0.5 m (gtsam_points_amd.synthetic.make_c2_workload; rank r uses seed 42 + r).
A "step" is one linearize() pass as the optimizer sees it.  N = 1: gp_vgicp_batch_linearize -- the pose rides in the kernel arguments, ONE launch (the
tile kernel's last workgroups finalize, --finalize two-kernel for tile kernel + finalize kernel), records into host memory, the host polls completion
words.  N > 1: pose -> tile kernel -> finalize kernel -> ONE exchange of the ranks' records over xGMI -> sync: by default every rank stores its record straight into every
peer's buffer and flags its arrival (one single-workgroup kernel, which also writes the complete [N x 122] f64 stack to pinned host memory: csrc/gp_peer.hip; validated
before use, all ranks fall back together); --exchange all_gather / all_reduce: ONE RCCL collective over the stacked record buffer (in place / zeroed stack + sum) -> D2H.
Inputs (source cloud, voxel map) are resident in HBM before the timed region.  value = N * 1e6 * K / elapsed.
Weak scaling: per-GPU work is fixed as N grows.

Extra objects on the JSON line:
  roofline     -- dominant kernel (vgicp_stream_kernel): algorithmic bytes (SURVEY.md 8(d):
                  48 N_src + 16 N_buckets + 52 N_voxels + 560) / its mean duration AS THE TIMED STEPS RAN IT: the fused kernel stamps
                  its own start, last partial row and hand-over on the device's 100 MHz clock (kernel_ms = streaming part, fused_kernel_ms =
                  with the finalize tail).  Beside it: the same kernel back to back under HIP events on its launch stream
                  (kernel_ms_back_to_back: the device's sustained state) and the committed rocprofv3 per-dispatch figures of the driver's
                  command (rocprof_*).  Peak 8 TB/s HBM3E.
  cpu_baseline -- the reference's own CPU factor (oracle/_ref/libref.so, kind "reference"; the C restatement, kind "port", when
                  that library is absent) timed on this box's cores on the same workload, rank 0 / N=1 only.
  c4           -- BASELINE configs[3] next to the headline: the 4096-factor graph (512 submaps x 32768 points, 8 factors per
                  source submap, 1.0 m voxels) partitioned over the N ranks by source submap with the target maps a shard
                  references replicated onto it (gtsam_points_amd.synthetic.c4_factor_pairs / make_c4_submaps, plan from
                  gp_shard_plan_create); one step = every rank's batched linearise into its rows of the zeroed [4096 x 122] f64
                  stack + ONE all-reduce (RCCL) + D2H.  Strong scaling: total work is fixed as N grows.  --no-c4 skips it.
  configs      -- the remaining BASELINE configs under the driver's clock (rank 0, N = 1; --no-configs skips them): C1 the two full
                  data/kitti_00 scans @0.5 m (covariances from gp_estimate_covariances), C3 the 256-factor submap graph as ONE batched
                  call through gp_vgicp_batch_linearize_view, C5 k-NN covariance estimation + GICP linearise at 1 M points.  Each
                  with ms, corr/s (points/s), a roofline object for its dominant kernel, parity against and the time of the
                  REFERENCE's own CPU code (oracle/_ref/libref.so; the C restatement when that is absent).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable)


class PhaseGuard:
    """Time-box of a phase (VERDICT r04 #5d): a rank that hangs in a collective or a rendezvous must fail in about two minutes, not sit on the lease.  A timer thread
    that finds the phase still open says which one on stderr and ends the PROCESS (os._exit: a hung RCCL call cannot be interrupted from Python); torchrun then tears
    the other ranks down."""

    def __init__(self, seconds, name):
        self.seconds, self.name, self._timer = float(seconds), name, None

    def __enter__(self):
        import threading

        def expire():
            sys.stderr.write(json.dumps(dict(error=f"bench.py: phase '{self.name}' exceeded its {self.seconds:.0f} s time box on rank {os.environ.get('RANK', '0')}; aborting")) + "\n")
            sys.stderr.flush()
            os._exit(124)

        self._timer = threading.Timer(self.seconds, expire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False


KERNEL_NAMES = {
    12: "vgicp_stream_kernel<linearise, non-temporal source stream, in-argument descriptor> (gp_vgicp_stream.hpp): 1024 workgroups, balanced chunk plan",
    8: "vgicp_pipeline_kernel<look-ahead> (gp_vgicp_tile.hpp)",
    2: "vgicp_pipeline_kernel<hashed line table> (gp_vgicp_tile.hpp)",
}


def _effective_kernel(lib, batch):
    v = C.c_int(-1)
    lib.gp_vgicp_batch_get_tuning(batch, 6, C.byref(v))  # GP_TUNE_EFFECTIVE_KERNEL
    return v.value


def _load_split():
    """rocprofv3 per-dispatch durations of the tile kernel by launch pattern (scripts/kernel_trace_split.py over the driver's bench command, builder-run)"""
    path = os.path.join(ROOT, "profiles", "kernel_trace_split.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return dict(in_step_ms=round(d["in_step"]["mean_us"] * 1e-3, 5), all_ms=round(d["all"]["mean_us"] * 1e-3, 5),
                    fused_in_step_ms=round(d["fused_in_step"]["mean_us"] * 1e-3, 5) if "fused_in_step" in d else None,
                    source="profiles/kernel_trace_split.json: builder-run rocprofv3 --kernel-trace per-dispatch durations of `bench.py --steps 20 --warmup 5` (fused) and of the same "
                           "with --finalize two-kernel (in step / mean); NOT measured in this run -- the cross-check of kernel_ms")
    except Exception:
        return {}


def _load_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (profiles/), or None.
    bench.py cannot collect PMC counters itself; scripts/gpu_check.sh does, in separate rocprofv3 --pmc passes,
    and scripts/pmc_summary.py applies the calibration."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get("tile_kernel_hbm_bytes_per_launch"), f"profiles/hbm_traffic.json ({d.get('source', 'builder-run rocprofv3 --pmc passes')}); NOT measured in this run"
    except Exception:
        return None, None


def measure_traffic(args):
    """roofline.traffic measured IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only, as
    MI355X_MICROARCH.md's HBM section prescribes) over a small child run of this file (--pmc-child: 7 fused steps of the same workload + a calibration stream of known
    48 N bytes in the tile kernel's own access pattern), read side scaled on the calibration stream (rocprofv3's FETCH_SIZE prices a 128-B request at 64 B on gfx950:
    DESIGN.md 6).  Never raises; returns {} / {"error": ...} when rocprofv3 is absent or a pass fails, and the committed figure is quoted instead."""
    import csv
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict

    exe = shutil.which("rocprofv3")
    if not exe:
        return dict(error="rocprofv3 not on PATH")
    means = {}
    try:
        t0 = time.time()
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(prefix="gp_pmc_", dir="/tmp") as tmp:
                cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                       "--source-points", str(args.source_points), "--target-points", str(args.target_points), "--resolution", str(args.resolution)]
                env = dict(os.environ, TMPDIR="/tmp")
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=150, cwd="/tmp", env=env)
                acc = defaultdict(list)
                for dirpath, _dirs, files in os.walk(tmp):
                    for fn in files:
                        if fn.endswith("counter_collection.csv"):
                            with open(os.path.join(dirpath, fn)) as f:
                                for row in csv.DictReader(f):
                                    if row.get("Counter_Name") == ctr:
                                        acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
                if not acc:
                    return dict(error=f"rocprofv3 --pmc {ctr}: no counter rows (exit code {p.returncode}): {p.stderr[-300:]}")
                means[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

        def find(d, needle):
            for k, v in d.items():
                if needle in k:
                    return v
            return None

        calib, tile_f, tile_w = find(means["FETCH_SIZE"], "calibration_stream_kernel"), find(means["FETCH_SIZE"], "vgicp_stream_kernel"), find(means["WRITE_SIZE"], "vgicp_stream_kernel")
        if not calib or not tile_f:
            return dict(error="the profiled child ran no calibration / stream kernel")
        scale = 48.0 * args.source_points / (calib[0] * 1024.0)
        return dict(tile_kernel_hbm_bytes_per_launch=int(tile_f[0] * 1024.0 * scale + (tile_w[0] if tile_w else 0.0) * 1024.0), fetch_size_kib=round(tile_f[0], 1),
                    write_size_kib=round(tile_w[0], 1) if tile_w else None, launches=tile_f[1], calibration_fetch_kib=round(calib[0], 1), fetch_scale=round(scale, 4),
                    seconds=round(time.time() - t0, 1),
                    source="measured in THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, --kernel-trace only) over a child run of 7 fused steps of the same "
                           "workload; read side scaled on a calibration stream of known bytes (fetch_scale); per launch of vgicp_stream_kernel")
    except Exception as exc:
        return dict(error=f"{type(exc).__name__}: {exc}")


def run_c4_inlib(lib, gpa, _capi, synthetic, torch, home_device, steps, max_devices=0):
    """The same 4096-factor configuration through the IN-LIBRARY sharded path a C++ optimizer process would use
    (gp_vgicp_multi_batch_*: ONE process drives every visible device, ncclCommInitAll, one ncclAllReduce of the [4096 x 122] f64
    stack per linearise; replaces the per-factor loop of src/gtsam_points/cuda/nonlinear_factor_set_gpu.cpp:64-139).  Only run
    when the process sees more than one device; rank 0 only.  Returns a dict (never raises: an error is reported as a string)."""
    from gtsam_points_amd.distributed import MultiDeviceBatch, partition_factors

    ndev = torch.cuda.device_count()
    if max_devices > 0:
        ndev = min(ndev, max_devices)
    try:
        t_setup = time.time()
        pairs = synthetic.c4_factor_pairs()
        F = len(pairs)
        parts = partition_factors([synthetic.C4_POINTS] * F, ndev)
        sub = synthetic.make_c4_submaps(range(synthetic.C4_SUBMAPS))
        factors, keep = [], []
        for dev, (b, e) in enumerate(parts):
            torch.cuda.set_device(dev)
            _capi.check(lib.gp_set_device(dev), "gp_set_device")
            mine = pairs[b:e]
            clouds = {i: gpa.PointCloudGPU(sub[i][0], sub[i][1], device=f"cuda:{dev}") for i in sorted({i for p in mine for i in p})}
            maps = {}
            for t in sorted({t for t, _ in mine}):
                m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
                m.insert(clouds[t])
                maps[t] = m
            factors += [gpa.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s]) for t, s in mine]
            keep.append((clouds, maps))
        torch.cuda.set_device(home_device)
        _capi.check(lib.gp_set_device(home_device.index), "gp_set_device")
        poses = np.stack([np.ascontiguousarray(synthetic.c4_delta(sub, t, s).T).reshape(16) for t, s in pairs]).copy()
        out = np.zeros((F, 122))
        t_setup = time.time() - t_setup
        res = dict(devices=ndev, unit="point-correspondences/s", setup_s=round(t_setup, 1),
                   note="host wall per gp_vgicp_multi_batch_linearize (poses in host memory -> all 4096 records in host memory); compute / exchange from the library's own HIP events; "
                        "one leg per exchange: in-place ncclAllGather of the equal contiguous shards, ncclAllReduce of the zeroed stack, and no collective (every shard's finalize "
                        "kernel stores its records straight into the one host-pinned stack)")
        ref = None
        for use_rccl, leg in [(2, "all_gather"), (1, "all_reduce"), (0, "no_collective")]:
            mb = MultiDeviceBatch(factors, use_rccl=use_rccl)
            for _ in range(3):
                mb.linearize_flat(poses, out)
            comp, exch = [], []
            t0 = time.perf_counter()
            for _ in range(steps):
                mb.linearize_flat(poses, out)
                tm = mb.last_timing()
                comp.append(tm["ms_compute"])
                exch.append(tm["ms_exchange"])
            ms = (time.perf_counter() - t0) / steps * 1e3
            if ref is None:
                ref = out.copy()
            res[leg] = dict(exchange=mb.exchange, shards=mb.num_shards, ms=round(ms, 4), compute_ms=round(float(np.median(comp)), 4), exchange_ms=round(float(np.median(exch)), 4),
                            value=round(F * synthetic.C4_POINTS / (ms * 1e-3), 1), records_equal_first_leg=bool(np.array_equal(ref, out)))
            del mb
        res["inlier_fraction"] = round(float(ref[:, 0].sum()) / (F * synthetic.C4_POINTS), 4)
        del factors, keep
        return res
    except Exception as exc:  # the headline must survive a failure of this optional leg
        try:
            torch.cuda.set_device(home_device)
            lib.gp_set_device(home_device.index)
        except Exception:
            pass
        return dict(devices=ndev, error=f"{type(exc).__name__}: {exc}")


def run_c4(args, lib, gpa, _capi, synthetic, torch, dist, rank, world, device, stream, dist_on=False):
    """BASELINE configs[3]: 4096 pairwise factors sharded over the ranks (see the module docstring).  Returns the `c4` object
    (rank 0) or None."""
    from gtsam_points_amd.distributed import RECORD_DOUBLES, ShardedLinearizer, partition_factors

    t_setup = time.time()
    pairs = synthetic.c4_factor_pairs()
    F = len(pairs)
    begin, end = partition_factors([synthetic.C4_POINTS] * F, world)[rank]
    mine = pairs[begin:end]
    need = sorted({i for p in mine for i in p})
    sub = synthetic.make_c4_submaps(need)
    clouds, maps = {}, {}
    for i in need:
        clouds[i] = gpa.PointCloudGPU(sub[i][0], sub[i][1], device=device)
    for t in sorted({t for t, _ in mine}):
        m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        m.insert(clouds[t])
        maps[t] = m
    sptr = C.c_void_p(stream.cuda_stream)
    factors = [gpa.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s], stream=sptr) for t, s in mine]
    n_local = len(factors)
    arr = (C.c_void_p * max(n_local, 1))(*[f._h.value for f in factors])
    batch = C.c_void_p()
    _capi.check(lib.gp_vgicp_batch_create(arr, n_local, sptr, C.byref(batch)), "gp_vgicp_batch_create")
    deltas = [synthetic.c4_delta(sub, t, s) for t, s in mine]
    poses = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in deltas]).copy() if n_local else np.zeros((0, 16))
    t_setup = time.time() - t_setup

    def issue(poses_local, view):
        _capi.check(lib.gp_vgicp_batch_issue_linearize(batch, poses_local.ctypes.data, C.c_void_p(view.data_ptr())), "gp_vgicp_batch_issue_linearize")

    sharded = ShardedLinearizer(F, (begin, end), device, issue, always_exchange=dist_on, exchange=args.c4_exchange)
    host_out = torch.zeros((F, RECORD_DOUBLES), dtype=torch.float64).pin_memory()

    def step():
        stacked = sharded.linearize(poses)
        host_out.copy_(stacked, non_blocking=True)
        stream.synchronize()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if dist_on:  # set-up is host work of uneven length (casting the submaps): meet first, so that the box below times collectives only
        with PhaseGuard(600.0, "c4 set-up rendezvous"):
            dist.barrier()
    guard = PhaseGuard(args.phase_seconds if dist_on else 900.0, "c4 steps and exchange")
    with guard:
        for _ in range(3):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.c4_steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        # the exchange alone, both forms: zeroing + all-reduce of the stacked records, and the in-place all-gather (when the plan qualifies); HIP events on the stream they are issued on
        ar_ms, ag_ms = 0.0, None
        if dist_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            e0.record(stream)
            for _ in range(10):
                sharded.stacked.zero_()
                dist.all_reduce(sharded.stacked, op=dist.ReduceOp.SUM)
            e1.record(stream)
            e1.synchronize()
            ar_ms = e0.elapsed_time(e1) / 10
            if sharded.exchange == "all_gather":
                try:
                    barrier()
                    e0.record(stream)
                    for _ in range(10):
                        dist.all_gather_into_tensor(sharded.stacked, sharded.own_rows)
                    e1.record(stream)
                    e1.synchronize()
                    ag_ms = e0.elapsed_time(e1) / 10
                except (RuntimeError, ValueError, NotImplementedError):
                    ag_ms = None
    ms_total, ms_main, ms_fin = C.c_float(), C.c_float(), C.c_float()
    alg = 0
    if n_local:
        _capi.check(lib.gp_vgicp_batch_time_linearize(batch, poses.ctypes.data, 10, C.byref(ms_total), C.byref(ms_main), C.byref(ms_fin)), "time_linearize")
        alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
    stats = torch.tensor([elapsed, ms_main.value, float(alg), float(n_local)], dtype=torch.float64, device=device)
    if dist_on:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    else:
        mx, sm = stats, stats
    elapsed_max, tile_ms_max, alg_sum = float(mx[0]), float(mx[1]), float(sm[2])
    inliers = float(host_out[:, 0].sum())
    lib.gp_vgicp_batch_destroy(batch)
    del factors, maps, clouds
    inlib = None
    if rank == 0 and world == 1 and not dist_on and torch.cuda.device_count() > 1 and not args.no_c4_inlib:
        # ONE process driving every visible device: in a process of its own with a time limit -- this leg has never run on more than one device (the builder's boxes
        # have one), and neither a hang nor a crash of it may take the headline line with it.  Only in the single-process run (N = 1 on a multi-GPU node): under
        # torch.distributed the other ranks own those devices.
        import subprocess

        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--c4-inlib-only", "--c4-steps", str(args.c4_steps)], capture_output=True, text=True, timeout=300)
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            inlib = json.loads(lines[-1]) if lines else dict(error=f"no result (exit code {p.returncode}): {p.stderr[-400:]}")
        except subprocess.TimeoutExpired:
            inlib = dict(error="the in-library multi-device leg did not finish within 300 s and was stopped")
        except Exception as exc:
            inlib = dict(error=f"{type(exc).__name__}: {exc}")
    if rank != 0:
        return None
    points = F * synthetic.C4_POINTS
    ms = elapsed_max / args.c4_steps * 1e3
    return dict(
        workload="BASELINE configs[3]: 4096 pairwise VGICP factors (512 submaps x 32768 pts, 1.0 m voxels), sharded by source submap over the ranks",
        factors=F, points_per_linearize=points, n_gpus=world, scaling="strong", steps=args.c4_steps,
        ms_per_linearize=round(ms, 4), value=round(points / (ms * 1e-3), 1), unit="point-correspondences/s",
        exchange=sharded.exchange, allreduce_ms=round(ar_ms, 4), allgather_ms=round(ag_ms, 4) if ag_ms is not None else None, stack_bytes=F * RECORD_DOUBLES * 8,
        tile_kernel_ms_slowest_rank=round(tile_ms_max, 5), algorithmic_bytes_total=int(alg_sum),
        algorithmic_frac_per_gpu=round(alg_sum / world / (tile_ms_max * 1e-3) / 8e12, 4) if tile_ms_max > 0 else None,
        algorithmic_frac_note="algorithmic bytes (SURVEY.md 8(d)) charge every factor its own 48 B/pt source stream although each source cloud serves 8 factors "
                              "(unique data ~0.9 GB of 7.9 GB) and ~half of the points miss: NOT an HBM fraction, no roofline credit claimed",
        inlib=inlib,
        factors_rank0=n_local, inlier_fraction=round(inliers / points, 4), setup_s=round(t_setup, 1),
        step="per rank: batched tile + finalize kernels into own rows of the [4096 x 122] f64 stack -> ONE collective (RCCL; `exchange`: in-place all-gather of the equal "
             "contiguous shards, or zeroed stack + all-reduce) -> D2H -> sync",
    )


BLOCKS = ["H_target", "H_source", "H_target_source", "b_target", "b_source"]


def pick_cpu_threads(avail, make, run, reps=3):
    """The CPU baseline is the reference's code on THIS box's cores, at the thread count that serves it best: a short probe over {all, 1/2, 1/4, 32, 16} threads.  The
    container may see more hardware threads than its CPU quota gives it, and a small factor does not scale to hundreds of threads: on one box of round 5 a 22 k-point
    factor took 228 ms with the 256 threads omp_get_max_threads() reported and 1.4 ms with 16.  make(threads) -> object, run(object) = one timed pass."""
    timed = []
    for c in sorted({avail, max(avail // 2, 1), max(avail // 4, 1), min(avail, 32), min(avail, 16)}):
        o = make(c)
        run(o)  # warm-up
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            run(o)
            ts.append(time.perf_counter() - t)
        timed.append((c, float(np.median(ts))))
    fastest = min(t for _, t in timed)
    return next(c for c, t in timed if t <= 1.15 * fastest)  # (ascending counts: of those within 15 % of the fastest, the one with the fewest threads -- the steadiest)


def run_lm_config(workload, gpa, gpu_factors, cpu_factors, pairs, num_poses, truth, values0, sptr, device, cores, kind):
    """configs.lm_*: the reference's LM cadence (bench_lm.py; levenberg_marquardt_ext.cpp:107-143,188-392) over a graph of VGICP factors -- per iteration host to host, by
    phase, on the GPU path (batched linearise, records stay in HBM, block-sparse LL^T on the device; and the same with a host-side numpy solve) and over the checker's CPU
    factors (the reference's own IntegratedVGICPFactor when oracle/_ref is built) as cpu_baseline.  truth None: the CPU run's result is the reference the GPU run is held to."""
    import bench_lm

    cg = bench_lm.CpuGraph(cpu_factors, pairs, num_poses, fixed=0)
    res_cpu = bench_lm.run_lm(cg, values0, max_iterations=30)
    gate_ref = truth if truth is not None else res_cpu["values"]
    cpu = bench_lm.summarize(res_cpu, cg, gate_ref, "cpu")
    out = dict(workload=workload, cadence="linearize(values) -> [solve (A + lambda I) dx = b -> retract -> error(new values) on the linearisation's correspondences] until accepted; "
               "lambda 1e-5, x10 / /10, minModelFidelity 1e-3, relativeErrorTol 1e-5 (GTSAM defaults; levenberg_marquardt_ext.cpp:188-392)",
               gate="max over poses, relative to the fixed pose: rotation < 0.015 rad, translation < 0.15 m (test_matching_cost_factors.cpp:227) against "
               + ("the generator's ground truth" if truth is not None else "the CPU run's result (real scans: no ground truth)"))
    for solver in ("device", "device-three-calls", "host"):  # device = the damped build + solve as one call (gp_*_system_step); -three-calls = round 4's build / download / solve
        gg = bench_lm.GpuGraph(gpa, gpu_factors, pairs, num_poses, fixed=0, solver=solver, stream=sptr, device=device)
        bench_lm.run_lm(gg, values0, max_iterations=30)  # warm-up: first-use table builds, allocations
        best = None
        for _ in range(3):
            r = bench_lm.run_lm(gg, values0, max_iterations=30)
            if best is None or r["seconds"] < best["seconds"]:
                best = r
        obj = bench_lm.summarize(best, gg, gate_ref, f"gpu, {solver} solve")
        gg.sync_phases = True
        split = bench_lm.summarize(bench_lm.run_lm(gg, values0, max_iterations=30), gg, gate_ref, "split")
        obj["ms_per_iteration_by_phase"] = split["ms_per_iteration_by_phase"]
        obj["dominant_phase"] = split["dominant_phase"]
        obj["phase_note"] = ("phases from a run that waits for the linearise before the solve is issued (the un-synchronised run queues the solver's kernels behind it: its "
                             "ms_per_iteration is the figure of merit); glue = numpy pose algebra of the harness (relative poses, retract), not library time")
        obj["pose_vs_cpu_run"] = dict(zip(("rotation_rad", "translation_m"), [round(max(x), 6) for x in zip(*[bench_lm.pose_error(best["values"][k], res_cpu["values"][k]) for k in range(num_poses)])]))
        gg.close()
        out[{"device": "gpu_device_solve", "device-three-calls": "gpu_device_solve_three_calls", "host": "gpu_host_solve"}[solver]] = obj
    cpu.update(cores=cores, kind=kind, sample=f"the whole loop once: every factor linearised / evaluated in turn with {cores} threads (the count a probe chose, pick_cpu_threads), numpy dense solve")
    out["cpu_baseline"] = cpu
    out["speedup_per_iteration"] = round(cpu["ms_per_iteration"] / out["gpu_device_solve"]["ms_per_iteration"], 1)
    return out


def _parity(L, Lo):
    out = {k: float(np.linalg.norm(getattr(L, k) - getattr(Lo, k)) / max(np.linalg.norm(getattr(Lo, k)), 1e-300)) for k in BLOCKS}
    out["error"] = float(abs(L.error - Lo.error) / max(abs(Lo.error), 1e-300))
    out["num_inliers_equal"] = bool(L.num_inliers == Lo.num_inliers)
    return out


def _median_ms(call, iters):
    ts = []
    for _ in range(iters):
        t = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3


def run_configs(args, lib, gpa, _capi, synthetic, torch, device, stream):
    """BASELINE configs[0], [2], [4] (C1, C3, C5 of SURVEY.md 8(d)) under the driver's clock.  GPU side = the product's synchronous entry
    points; CPU side = the reference's own code (oracle/_ref/libref.so) on all host cores, on a bounded sample, as checker and baseline."""
    import oracle  # checker / baseline only
    from oracle import refcapi

    use_ref = refcapi.available()
    avail = oracle.max_threads()
    kind = "reference" if use_ref else "port"
    VoxelMap = refcapi.RefVoxelMap if use_ref else oracle.OracleVoxelMap
    VGICP = refcapi.RefVGICPFactor if use_ref else oracle.OracleVGICPFactor
    GICP = refcapi.RefGICPFactor if use_ref else oracle.OracleGICPFactor
    sptr = C.c_void_p(stream.cuda_stream)
    out = {}

    def time_batch(factors, poses, iters, view=True):
        F = len(factors)
        arr = (C.c_void_p * F)(*[f._h.value for f in factors])
        batch = C.c_void_p()
        _capi.check(lib.gp_vgicp_batch_create(arr, F, sptr, C.byref(batch)), "gp_vgicp_batch_create")
        recs = np.zeros((F, _capi.LINEARIZED6_DOUBLES))
        vptr = C.c_void_p()
        pp, rp = C.c_void_p(poses.ctypes.data), C.c_void_p(recs.ctypes.data)
        for _ in range(5):
            _capi.check(lib.gp_vgicp_batch_linearize(batch, pp, rp), "gp_vgicp_batch_linearize")
        ms_copy = _median_ms(lambda: lib.gp_vgicp_batch_linearize(batch, pp, rp), iters)
        ms_view = _median_ms(lambda: lib.gp_vgicp_batch_linearize_view(batch, pp, C.byref(vptr)), iters)
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pp, min(iters, 50), C.byref(a), C.byref(b), C.byref(c)), "time_linearize")
        alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
        npts = int(lib.gp_vgicp_batch_total_points(batch))
        lib.gp_vgicp_batch_destroy(batch)
        roof = dict(bound="hbm", kernel="vgicp_stream_kernel (batched tile table)" if F > 1 else "vgicp_stream_kernel (in-argument descriptor)",
                    achieved=round(alg / (b.value * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg / (b.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    algorithmic_bytes=alg, kernel_ms=round(b.value, 5), finalize_kernel_ms=round(c.value, 5), device_pass_ms=round(a.value, 5), traffic=None)
        return recs, ms_copy, ms_view, npts, roof

    # ---- C1: the two full data/kitti_00 scans (shipped as tests/golden/kitti_00/*.bin), 0.5 m voxels, single linearise ----
    gdir = os.path.join(ROOT, "tests", "golden", "kitti_00")
    if os.path.exists(os.path.join(gdir, "000000.bin")):
        tp = np.fromfile(os.path.join(gdir, "000000.bin"), dtype=np.float32).reshape(-1, 3)
        sp = np.fromfile(os.path.join(gdir, "000001.bin"), dtype=np.float32).reshape(-1, 3)
        tgt, src = gpa.PointCloudGPU(tp, device=device), gpa.PointCloudGPU(sp, device=device)
        gpa.estimate_covariances_gpu(tgt, 10)
        gpa.estimate_covariances_gpu(src, 10)
        kitti_cov_ts = []
        for fr in (tgt, src, tgt, src, tgt, src, tgt):  # (alternating clouds, as for C5 below: a call does not find its own scratch arrays waiting)
            torch.cuda.synchronize()
            t = time.perf_counter()
            gpa.estimate_covariances_gpu(fr, 10)
            if fr is tgt:
                kitti_cov_ts.append(time.perf_counter() - t)
        kitti_cov_ms = float(np.median(kitti_cov_ts)) * 1e3
        vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
        vm.insert(tgt)
        f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src, stream=sptr)
        delta = synthetic.expmap(synthetic.C1B_PERTURBATION)
        pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
        recs, ms_copy, ms_view, npts, roof = time_batch([f], pose, 200)
        roof["note"] = "launch-bound: 124,605 points are 487 workgroups of 256 points; the kernel is a few microseconds whatever its bytes"
        tc, sc = tgt.download("covs"), src.download("covs")  # float32 exactly as the kernels read them
        om = VoxelMap(0.5)
        om.insert(tp, tc)
        cores = pick_cpu_threads(avail, lambda c: VGICP(om, sp, sc, c), lambda o: o.linearize(delta))
        fo = VGICP(om, sp, sc, cores)
        Lo = fo.linearize(delta)
        cpu_ms = _median_ms(lambda: fo.linearize(delta), 10)
        f1 = VGICP(om, sp, sc, 1)
        cpu1_ms = _median_ms(lambda: f1.linearize(delta), 3)
        out["C1"] = dict(
            workload="BASELINE configs[0]: two full data/kitti_00 scans (124,668 / 124,605 pts), covariances k=10 from gp_estimate_covariances, 0.5 m voxels, single linearise",
            points=npts, num_voxels=vm.voxelmap_info.num_voxels, ms=round(ms_copy, 5), ms_view=round(ms_view, 5), corr_per_s=round(npts / ms_copy * 1e3, 1), roofline=roof,
            covariances_ms=round(kitti_cov_ms, 4),
            cpu_baseline=dict(value=round(npts / cpu_ms * 1e3, 1), unit="point-correspondences/s", cores=cores, cores_available=avail, kind=kind, ms=round(cpu_ms, 3), ms_1thread=round(cpu1_ms, 3),
                              sample="10 full linearize() passes of the same factor (the reference's default is 1 thread: ms_1thread)"),
            parity_vs_reference=_parity(gpa.LinearizedSystem6.from_doubles(recs[0]), Lo), inlier_fraction=round(float(recs[0, 0]) / npts, 4))
        if not args.no_lm:
            try:
                out["lm_c1"] = run_lm_config("BASELINE configs[0] as an optimisation: scan 000001 registered to the map of scan 000000 from the identity (one factor, one free pose)",
                                             gpa, [f], [fo], [(0, 1)], 2, None, np.stack([np.eye(4), np.eye(4)]), sptr, device, cores, kind)
            except Exception as exc:  # the headline must survive an optional leg
                out["lm_c1"] = dict(error=f"{type(exc).__name__}: {exc}")
        del f, vm, tgt, src

    # ---- C3: 256-factor submap graph, ONE batched call ----
    t0 = time.time()
    g = synthetic.make_c3_graph()
    clouds = [gpa.PointCloudGPU(p, c, device=device) for p, c in g["clouds"]]
    maps = []
    for c in clouds:
        m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        m.insert(c)
        maps.append(m)
    factors = [gpa.IntegratedVGICPFactorGPU(t, s_, maps[t], clouds[s_], stream=sptr) for t, s_ in g["pairs"]]
    poses = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in g["deltas"]]).copy()
    t_setup = time.time() - t0
    recs, ms_copy, ms_view, npts, roof = time_batch(factors, poses, 50)
    roof["note"] = ("algorithmic bytes charge every factor its own source cloud (SURVEY.md 8(d)); four factors share each cloud and the re-reads hit L2, "
                    "so this fraction is not an HBM fraction")
    sample = list(range(0, len(factors), 8))  # every 8th factor: 32 reference linearisations
    omaps, worst, t_cpu = {}, 0.0, 0.0
    t0_, s0_ = g["pairs"][sample[0]]
    omaps[t0_] = VoxelMap(1.0)
    omaps[t0_].insert(*g["clouds"][t0_])
    cores = pick_cpu_threads(avail, lambda c: VGICP(omaps[t0_], g["clouds"][s0_][0], g["clouds"][s0_][1], c), lambda o: o.linearize(g["deltas"][sample[0]]))
    for k in sample:
        t, s_ = g["pairs"][k]
        if t not in omaps:
            omaps[t] = VoxelMap(1.0)
            omaps[t].insert(*g["clouds"][t])
        fo = VGICP(omaps[t], g["clouds"][s_][0], g["clouds"][s_][1], cores)
        fo.linearize(g["deltas"][k])
        reps = []
        for _ in range(3):  # (median of three: the first pass behind other host work pays for waking the team)
            tt = time.perf_counter()
            Lo = fo.linearize(g["deltas"][k])
            reps.append(time.perf_counter() - tt)
        t_cpu += float(np.median(reps))
        par = _parity(gpa.LinearizedSystem6.from_doubles(recs[k]), Lo)
        worst = max(worst, max(par[b] for b in BLOCKS), par["error"])
        assert par["num_inliers_equal"], k
    cpu_ms_graph = t_cpu / len(sample) * len(factors) * 1e3
    out["C3"] = dict(
        workload="BASELINE configs[2]: 256-factor submap graph (64 submaps x ~22k pts, factors i -> i+1..i+4 and back, 1.0 m voxels), ONE batched linearise "
                 "through gp_vgicp_batch_linearize_view",
        factors=len(factors), points=npts, ms=round(ms_view, 5), ms_with_copy=round(ms_copy, 5), corr_per_s=round(npts / ms_view * 1e3, 1), roofline=roof,
        cpu_baseline=dict(value=round(npts / cpu_ms_graph * 1e3, 1), unit="point-correspondences/s", cores=cores, cores_available=avail, kind=kind, ms=round(cpu_ms_graph, 2),
                          sample=f"{len(sample)} of the 256 factors (every 8th), the median of three linearize() passes each after a warm-up, {cores} threads per factor, sequential over factors "
                                 "as graph_.linearize does; scaled x8"),
        parity_vs_reference_max=worst, parity_factors_checked=len(sample), inlier_fraction=round(float(recs[:, 0].sum()) / npts, 4), setup_s=round(t_setup, 1))
    if not args.no_lm:
        try:
            n_sub = len(g["clouds"])
            for t in range(n_sub):
                if t not in omaps and any(p[0] == t for p in g["pairs"]):
                    omaps[t] = VoxelMap(1.0)
                    omaps[t].insert(*g["clouds"][t])
            cpu_factors = [VGICP(omaps[t], g["clouds"][s_][0], g["clouds"][s_][1], cores) for t, s_ in g["pairs"]]
            truth = np.stack(g["stations"][:n_sub])
            import bench_lm

            v0 = truth @ bench_lm.expmap_many(np.random.default_rng(8191).uniform(-0.1, 0.1, (n_sub, 6)))  # ground truth o Expmap(U(-0.1, 0.1)^6), seed 8191: the reference tests' noise
            v0[0] = truth[0]
            out["lm_c3"] = run_lm_config("BASELINE configs[2] as an optimisation: the 256-factor / 64-submap graph from ground truth o Expmap(U(-0.1, 0.1)^6) (seed 8191), pose 0 held",
                                         gpa, factors, cpu_factors, g["pairs"], n_sub, truth, v0, sptr, device, cores, kind)
            del cpu_factors
        except Exception as exc:
            out["lm_c3"] = dict(error=f"{type(exc).__name__}: {exc}")
    del factors, maps, clouds

    # ---- C5: k-NN covariance estimation (k = 10) + IntegratedGICPFactor linearise, 1 M points ----
    d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
    tgt, src = gpa.PointCloudGPU(d["target_points"], device=device), gpa.PointCloudGPU(d["source_points"], device=device)
    torch.cuda.synchronize()
    for fr in (tgt, src, tgt, src, tgt, src):  # (warm-up: the first calls behind another phase pay for their scratch blocks, gp_host.hpp BlockCache)
        gpa.estimate_covariances_gpu(fr, 10)
    # the config's cloud is the SOURCE cloud (the CPU baseline and the parity check run on it); the target cloud of the same scene (a denser, map-like sampling whose
    # search takes about twice as long) is timed beside it, and the two alternate so that neither call finds the other's scratch arrays waiting
    ts, ts_tgt = [], []
    for fr in (tgt, src, src, tgt, src, src, tgt, src, src, src):
        torch.cuda.synchronize()
        t = time.perf_counter()
        n_short = gpa.estimate_covariances_gpu(fr, 10)
        (ts if fr is src else ts_tgt).append(time.perf_counter() - t)
        if fr is src:
            short = n_short
    cov_ms = float(np.median(ts)) * 1e3
    cov_tgt_ms = float(np.median(ts_tgt)) * 1e3
    kt = gpa.features.covariance_kernel_times(src, 10) if hasattr(gpa.features, "covariance_kernel_times") else None
    side = None
    try:  # which of its candidate side streams the covariance call uses beside the stream it was called on (the null stream here), and what the pipe probe measured for each
        delays, chosen = (C.c_float * 4)(), C.c_int(-1)
        _capi.check(lib.gp_debug_side_stream_probe(None, delays, C.byref(chosen)), "gp_debug_side_stream_probe")
        side = dict(probe_delay_us=[round(float(x), 1) for x in delays], chosen=chosen.value,
                    note="delay between the first workgroup of a device-filling grid on the caller's stream and a wave on the candidate stream: ~1 us = another dispatch pipe, "
                         "tens of us = the same pipe (the second covariance launch would start when the first is fully placed: +0.07 ms per call, DESIGN.md 4.8)")
    except Exception as exc:
        side = dict(error=f"{type(exc).__name__}: {exc}")
    got = src.download("covs").astype(np.float64)
    cov_fn = refcapi.ref_estimate_covariances if use_ref else (lambda p, k_, c: oracle.estimate_covariances(p, k_, c)[0])
    cores = pick_cpu_threads(avail, lambda c: c, lambda c: cov_fn(d["source_points"][:100_000], 10, c))  # (probe on a tenth of the cloud)
    if use_ref:
        t = time.perf_counter()
        ref_cov = refcapi.ref_estimate_covariances(d["source_points"], 10, cores)
        cov_cpu_ms = (time.perf_counter() - t) * 1e3
    else:
        t = time.perf_counter()
        ref_cov, _ = oracle.estimate_covariances(d["source_points"], 10, cores)
        cov_cpu_ms = (time.perf_counter() - t) * 1e3
    rel = np.linalg.norm((got - ref_cov).reshape(len(got), -1), axis=1) / np.linalg.norm(ref_cov.reshape(len(got), -1), axis=1)
    fg = gpa.IntegratedGICPFactorGPU(0, 1, tgt, src)
    delta5 = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    fg.linearize_delta(delta5)
    gicp_ms = _median_ms(lambda: fg.linearize_delta(delta5), 20)
    L = fg.linearize_delta(delta5)
    tc, sc = tgt.download("covs"), src.download("covs")
    cores_cov = cores
    cores = pick_cpu_threads(avail, lambda c: GICP(d["target_points"], tc, d["source_points"], sc, c), lambda o: o.linearize(delta5))
    fo = GICP(d["target_points"], tc, d["source_points"], sc, cores)
    Lo = fo.linearize(delta5)
    gicp_cpu_ms = _median_ms(lambda: fo.linearize(delta5), 3)
    cov_roof = dict(bound="issue", kernel="covariance_kernel<10> (gp_knn.hip)", unit="ms",
                    note="not HBM-bound: the cloud (16 MB as float4) is re-read out of L1/L2; the stated bound is the vector-memory address path of the divergent per-lane "
                         "candidate gathers + the f64 insertions and the eigen-decomposition (DESIGN.md section 4.8)",
                    compulsory_bytes=48 * 1_000_000, hbm_frac_of_compulsory=round(48e6 / (cov_ms * 1e-3) / 8e12, 5))
    if kt:
        cov_roof.update(kt)
    out["C5"] = dict(
        workload="BASELINE configs[4]: k-NN covariance estimation (k=10, exact) + IntegratedGICPFactor linearise, 1 M source pts vs 1 M target pts",
        points=1_000_000,
        covariances=dict(ms=round(cov_ms, 4), ms_target_cloud=round(cov_tgt_ms, 4), ms_kitti_scan=(out.get("C1") or {}).get("covariances_ms"),
                         clouds_note="ms: the config's cloud (the 1 M-point C2 source); ms_target_cloud: the denser, map-like sampling of the same scene (1 M points); ms_kitti_scan: a real "
                                     "124,668-point scan (data/kitti_00/000000.bin), most of it far field -- round 4: 0.74 / 1.23-1.31 / 0.74 ms (profiles/r05_c5_ab.jsonl)",
                         points_per_s=round(1e6 / cov_ms * 1e3, 1), num_short=int(short), roofline=cov_roof, side_stream=side,
                         cpu_baseline=dict(value=round(1e6 / cov_cpu_ms * 1e3, 1), unit="points/s", cores=cores_cov, cores_available=avail, kind=kind, ms=round(cov_cpu_ms, 2),
                                           sample="one estimate_covariances pass over the same 1 M points (kd-tree build + 10-NN + eigen-regularisation; the 3x3 eigen-solver under the "
                                                  "reference code is the stand-in Jacobi iteration of oracle/ref_shim, not Eigen's closed form)"),
                         parity_vs_reference=dict(rel_err_median=float(np.median(rel)), frac_within_1e5=float((rel < 1e-5).mean()))),
        gicp=dict(ms=round(gicp_ms, 4), corr_per_s=round(1e6 / gicp_ms * 1e3, 1),
                  roofline=dict(bound="issue", kernel="gicp_correspond_kernel + gicp_tile_kernel<CORR> (gp_knn.hip)", unit="ms",
                                note="1-NN walk of the cell grid per point, then the VGICP algebra on the matched target point; arithmetic- and divergence-bound (DESIGN.md 4.8)",
                                compulsory_bytes=96 * 1_000_000, hbm_frac_of_compulsory=round(96e6 / (gicp_ms * 1e-3) / 8e12, 5)),
                  cpu_baseline=dict(value=round(1e6 / gicp_cpu_ms * 1e3, 1), unit="point-correspondences/s", cores=cores, cores_available=avail, kind=kind, ms=round(gicp_cpu_ms, 2),
                                    sample="3 linearize() passes (1-NN kd-tree search + evaluate) of the same factor"),
                  parity_vs_reference=_parity(L, Lo), inlier_fraction=round(L.num_inliers / 1e6, 4)))
    # ---- map build: the Gaussian voxel map of the 2 M-point C2 target at 0.5 m (replaces types/gaussian_voxelmap_gpu.cu:211-307), wall per gp_voxelmap_insert ----
    tgt2 = gpa.PointCloudGPU(d["target_points"], d["target_covs"], device=device) if len(d["target_points"]) >= 2_000_000 else None
    if tgt2 is None:
        d2 = synthetic.make_c2_workload(1000, 2_000_000, seed=42)
        tgt2 = gpa.PointCloudGPU(d2["target_points"], d2["target_covs"], device=device)
    ts, vmb = [], None
    for _ in range(25):
        vmb = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        vmb.insert(tgt2)
        ts.append(time.perf_counter() - t)
    mb_ms = float(np.median(ts[5:])) * 1e3
    nt = tgt2.size()
    out["map_build"] = dict(
        workload="GaussianVoxelMapGPU::insert of the 2 M-point C2 target cloud at 0.5 m (bit-reproducible binned build: bounding box, stable radix sort by (block, cell), cells, "
                 "occupancy-block grid, per-voxel statistics in f64, reference-visible bucket table)",
        points=nt, num_voxels=int(vmb.voxelmap_info.num_voxels), ms=round(mb_ms, 4), ms_min=round(float(np.min(ts[5:])) * 1e3, 4), points_per_s=round(nt / mb_ms * 1e3, 1),
        roofline=dict(bound="hbm", achieved=round(48.0 * nt / (mb_ms * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(48.0 * nt / (mb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                      algorithmic_bytes=48 * nt, traffic=None,
                      note="48 B per point read once (SURVEY.md 8(d), voxel-map build); host wall of the whole call (10 launches, three points where the host waits), not one kernel: the build is "
                           "a chain of latency-bound kernels at this size (DESIGN.md section 4.4)"))
    return out


def run_big_source(args, lib, gpa, _capi, synthetic, torch, device, stream, n_src=8_000_000):
    """The headline kernel on a source that does not fit the 256 MiB Infinity Cache (8 M points: 384 MB in the API layout, 288 MB packed), under the driver's clock: the
    only figure in the line that is DRAM bandwidth beyond doubt.  Same map, same kernel, same in-step stamps as the headline."""
    d = synthetic.make_c2_workload(n_src, 2_000_000, seed=42)
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"], device=device)
    src = gpa.PointCloudGPU(d["source_points"], d["source_covs"], device=device)
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    sptr = C.c_void_p(stream.cuda_stream)
    f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src, stream=sptr)
    arr = (C.c_void_p * 1)(f._h.value)
    batch = C.c_void_p()
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, sptr, C.byref(batch)), "gp_vgicp_batch_create")
    delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
    rec = np.zeros((1, _capi.LINEARIZED6_DOUBLES))
    pp, rp = C.c_void_p(pose.ctypes.data), C.c_void_p(rec.ctypes.data)
    t_wake = time.perf_counter()
    while time.perf_counter() - t_wake < 0.2:  # (device wake-up, as for the headline)
        _capi.check(lib.gp_vgicp_batch_linearize(batch, pp, rp), "gp_vgicp_batch_linearize")
    lib.gp_vgicp_batch_device_times(batch, 1, None, None, None)
    steps = 50
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.gp_vgicp_batch_linearize(batch, pp, rp)
    ms = (time.perf_counter() - t0) / steps * 1e3
    n_, su, ku = C.c_double(), C.c_double(), C.c_double()
    lib.gp_vgicp_batch_device_times(batch, 0, C.byref(n_), C.byref(su), C.byref(ku))
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pp, 20, C.byref(a), C.byref(b), C.byref(c)), "time_linearize")
    alg, act = int(lib.gp_vgicp_batch_algorithmic_bytes(batch)), int(lib.gp_vgicp_batch_actual_bytes(batch))
    lib.gp_vgicp_batch_destroy(batch)
    kms = su.value * 1e-3 if su.value > 0 else b.value
    return dict(workload=f"the headline factor with an {n_src // 1_000_000} M-point source (beyond the Infinity Cache), same 2 M-point map", points=n_src, steps=steps, ms_per_linearize=round(ms, 4),
                value=round(n_src / (ms * 1e-3), 1), unit="point-correspondences/s", inlier_fraction=round(float(rec[0, 0]) / n_src, 4),
                roofline=dict(bound="hbm", achieved=round(alg / (kms * 1e-3) / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              algorithmic_bytes=alg, actual_bytes=act, frac_actual=round(act / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), kernel_ms=round(kms, 5),
                              kernel_ms_source="the kernel's own 100 MHz stamps inside the timed steps (streaming part)" if su.value > 0 else "HIP events, back to back",
                              fused_kernel_ms=round(ku.value * 1e-3, 5), kernel_ms_back_to_back=round(b.value, 5), frac_back_to_back=round(alg / (b.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              traffic=None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--source-points", type=int, default=1_000_000)
    ap.add_argument("--target-points", type=int, default=2_000_000)
    ap.add_argument("--resolution", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--kernel-iters", type=int, default=50)
    ap.add_argument("--no-c4", action="store_true", help="skip the sharded 4096-factor configuration (BASELINE configs[3])")
    ap.add_argument("--c4-steps", type=int, default=30)
    ap.add_argument("--c4-exchange", choices=["all_gather", "all_reduce"], default="all_gather",
                    help="the c4 step's collective: all_gather = in place, half the bytes, no zeroing (falls back to the all-reduce when the shards are not equal contiguous ranges)")
    ap.add_argument("--exchange", choices=["peer", "all_gather", "all_reduce"], default="peer",
                    help="N > 1 headline step: how the ranks' records are exchanged.  peer (default) = every rank stores its record straight into every peer's buffer over xGMI and "
                         "flags its arrival, one single-workgroup kernel per step (csrc/gp_peer.hip; validated against known rows first, every rank falls back to all_gather together "
                         "when the buffers cannot be shared); all_gather = RCCL, in place, every row moved once; all_reduce = the north star's wording: sum over the zeroed stack")
    ap.add_argument("--no-c4-inlib", action="store_true", help="skip the single-process multi-device leg of c4 (run in a subprocess by the N = 1 run when it sees > 1 device)")
    ap.add_argument("--c4-inlib-only", action="store_true", help="(internal) run only the single-process multi-device leg of c4 and print its JSON object")
    ap.add_argument("--finalize", choices=["fused", "two-kernel"], default="fused",
                    help="synchronous step: fused = the library default (the last tile workgroups finalize, one launch); two-kernel = GP_TUNE_FUSED_FINALIZE 0")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1 / C3 / C5 objects (BASELINE configs[0], [2], [4])")
    ap.add_argument("--device-warmup-ms", type=float, default=300.0,
                    help="untimed: run the step for this long before the W warm-up steps, so that the timed steps see the device's settled power state (0 = off)")
    ap.add_argument("--no-big-source", action="store_true", help="skip the 8 M-point source (beyond the Infinity Cache) object")
    ap.add_argument("--no-mirror", action="store_true", help="A/B: stream the caller's 12 + 36 B per point instead of the packed 36-B mirror (GP_TUNE_SOURCE_MIRROR 0)")
    ap.add_argument("--phase-seconds", type=float, default=120.0, help="N > 1: time box of every distributed phase (rendezvous, warm-up, timed steps, c4): a hung rank ends the job")
    ap.add_argument("--no-cold", action="store_true", help="skip the ms_per_step_cold leg (the K timed steps without the device wake-up in front: round 3's protocol)")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic in this run (rocprofv3 --pmc passes of a small child run); the committed figure is quoted instead")
    ap.add_argument("--no-lm", action="store_true", help="skip configs.lm_c3 / lm_c1 (one Levenberg-Marquardt loop per graph, per-iteration cost by phase)")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) the small run the --pmc passes profile: 5 fused steps + the calibration stream, no output line")
    ap.add_argument("--inlib-devices", type=int, default=0, help="(internal) devices the --c4-inlib-only leg drives (0 = all visible)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

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
        print(json.dumps(run_c4_inlib(lib, gpa, _capi, synthetic, torch, torch.device("cuda:0"), max(args.c4_steps // 3, 5), args.inlib_devices)), flush=True)
        return
    dev_index = local_rank % torch.cuda.device_count()  # == local_rank on a full node; lets a 1-GPU box rehearse N > 1
    torch.cuda.set_device(dev_index)
    device = torch.device(f"cuda:{dev_index}")
    # GP_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, ShardedLinearizer, all-reduce of the record stack) with ONE rank -- how a
    # 1-GPU box runs the RCCL initialisation and a 1-rank ncclAllReduce that the gloo rehearsal cannot (tests/test_multi_gpu.py)
    dist_on = world > 1 or bool(os.environ.get("GP_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        import datetime

        backend = os.environ.get("GP_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; "gloo" only for the 1-GPU rehearsal
        box = datetime.timedelta(seconds=args.phase_seconds)  # the collectives' own time-out (the watchdog aborts the process); PhaseGuard is the belt to these braces
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
    factor = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src, stream=C.c_void_p(stream.cuda_stream))
    arr = (C.c_void_p * 1)(factor._h.value)
    batch = C.c_void_p()
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, C.c_void_p(stream.cuda_stream), C.byref(batch)), "gp_vgicp_batch_create")
    if args.finalize == "two-kernel":
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_FUSED_FINALIZE, 0), "finalize form")
    if args.no_mirror:
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_SOURCE_MIRROR, 0), "source mirror")
    delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()

    REC = _capi.LINEARIZED6_DOUBLES
    host_out = torch.zeros((world, REC), dtype=torch.float64).pin_memory()
    out_np = host_out.numpy()

    if not dist_on:
        # the product's synchronous entry point: pose in host memory -> records in host memory
        linearize = lib.gp_vgicp_batch_linearize  # bound once: the step is ~40 us, attribute lookups and .ctypes views are not free
        pose_ptr, out_ptr = C.c_void_p(pose.ctypes.data), C.c_void_p(out_np.ctypes.data)

        def step():
            if linearize(batch, pose_ptr, out_ptr) != 0:
                _capi.check(1, "gp_vgicp_batch_linearize")

    else:
        from gtsam_points_amd.distributed import ShardedLinearizer

        issue_linearize = lib.gp_vgicp_batch_issue_linearize
        pose_ptr = C.c_void_p(pose.ctypes.data)
        row_ptr = {}

        def issue(poses_local, view):  # (the row views are few objects -- one, or the peer exchange's two generations: their addresses are taken once)
            p = row_ptr.get(id(view))
            if p is None:
                p = row_ptr[id(view)] = (C.c_void_p(view.data_ptr()), view)  # (the view is kept: its id stays its own)
            if issue_linearize(batch, pose_ptr, p[0]) != 0:
                _capi.check(1, "gp_vgicp_batch_issue_linearize")

        sharded = ShardedLinearizer(world, (rank, rank + 1), device, issue, always_exchange=True, exchange=args.exchange, host_out=host_out)
        with PhaseGuard(args.phase_seconds, "exchange set-up"):  # (the first pass decides the exchange, collectively: buffers shared and validated, or the fall-back agreed)
            sharded.linearize(pose)
            torch.cuda.synchronize()
            sharded.check()
        delivers = sharded.delivers_to_host

        def step():
            # local kernels into the rank's row, then ONE exchange: direct stores into the peers' buffers over xGMI (the exchange kernel also fills host_out), or one RCCL
            # collective (in-place all-gather; --exchange all_reduce: zero + all-reduce) and a D2H copy
            stacked = sharded.linearize(pose)
            if not delivers:
                host_out.copy_(stacked, non_blocking=True)
            stream.synchronize()

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_child:  # the run the --pmc passes profile (measure_traffic): a few fused steps + the calibration stream of known bytes, nothing else
        for _ in range(7):
            step()
        barrier()
        _capi.check(_capi.load_tune().gp_debug_calibration_stream(src.ptr(src.points_gpu), src.ptr(src.covs_gpu), args.source_points, 5, C.c_void_p(stream.cuda_stream)), "calibration")
        torch.cuda.synchronize()
        lib.gp_vgicp_batch_destroy(batch)
        return None

    def timed_steps(label):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks"""
        with PhaseGuard(args.phase_seconds if dist_on else 600.0, label):
            for _ in range(args.warmup):
                step()
            barrier()
            lib.gp_vgicp_batch_device_times(batch, 1, None, None, None)  # reset: the kernel's own time stamps of the timed steps only
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            el = time.perf_counter() - t0
            if dist_on:
                sharded.check()  # (peer exchange: a peer that did not arrive within the kernel's time box is an error, not a stale stack)
            n_, su_, ku_ = C.c_double(), C.c_double(), C.c_double()
            lib.gp_vgicp_batch_device_times(batch, 0, C.byref(n_), C.byref(su_), C.byref(ku_))
            if dist_on:
                te = torch.tensor([el], dtype=torch.float64, device=device)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                el = float(te.item())
        return el, n_, su_, ku_

    # ms_per_step_cold: the same W + K protocol with NOTHING in front -- the device as seconds of host-side set-up left it, round 3's protocol --
    # so that rounds stay comparable whatever the wake-up below does
    cold = None
    if not args.no_cold:
        el_c, n_c, su_c, ku_c = timed_steps("timed steps (cold)")
        cold = dict(ms_per_step=round(el_c / args.steps * 1e3, 5), stream_us=round(su_c.value, 3) if n_c.value >= args.steps else None,
                    fused_kernel_us=round(ku_c.value, 3) if n_c.value >= args.steps else None)
    # device wake-up (untimed, before the W warm-up steps): seconds of host-side set-up leave the device in a low power state, and it takes ~10 ms of work before the
    # step settles -- scripts/r04_warm.py: 11.6-11.9 us for the first 400-600 steps behind 2 s of idle, 10.9-11.0 us from then on (profiles/r04_warm.jsonl).  An optimizer
    # loop runs in the settled state; the same synchronous step is run for --device-warmup-ms first
    t_wake, wake_steps = time.perf_counter(), 0
    with PhaseGuard(args.phase_seconds if dist_on else 600.0, "device wake-up"):
        if dist_on:
            # a step holds a collective: every rank must run the SAME number of them.  Chunks of 50 steps; after each the ranks agree (MAX over ranks of the time spent so
            # far: one tiny collective, untimed phase) whether the wake-up has lasted --device-warmup-ms -- a count alone (5000 steps at the ~60 us of an RCCL step) took
            # minutes with a slow backend (the gloo rehearsal on one GPU: profiles/r05_rehearsal_n*.log) and tripped the phase's time box
            spent = torch.zeros(1, dtype=torch.float64, device=device)
            while wake_steps < 5000 and args.device_warmup_ms > 0:
                for _ in range(50):
                    step()
                    wake_steps += 1
                    if wake_steps % 25 == 0:
                        torch.cuda.synchronize()
                spent[0] = (time.perf_counter() - t_wake) * 1e3
                dist.all_reduce(spent, op=dist.ReduceOp.MAX)
                if float(spent.item()) >= args.device_warmup_ms:
                    break
        else:
            while (time.perf_counter() - t_wake) * 1e3 < args.device_warmup_ms:
                step()
                wake_steps += 1
                if wake_steps % 25 == 0:
                    torch.cuda.synchronize()  # (the timed region is bracketed by device synchronisations: the wake-up runs the same pattern)
    elapsed, dev_steps, dev_stream_us, dev_kernel_us = timed_steps("timed steps")
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.source_points * args.steps / elapsed

    # ---- dominant-kernel roofline ----
    # (1) as the timed steps ran it: the fused kernel stamps its own start / last row in / sums out on the device's 100 MHz clock (gp_vgicp_batch_device_times)
    # (2) back to back: HIP events on the launch stream over a loop of tile-kernel launches (two-kernel form: the streaming part alone)
    ms_total, ms_main, ms_fin = C.c_float(), C.c_float(), C.c_float()
    _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, args.kernel_iters, C.byref(ms_total), C.byref(ms_main), C.byref(ms_fin)), "time_linearize")
    alg_bytes = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
    actual_bytes = int(lib.gp_vgicp_batch_actual_bytes(batch))
    mirrored = C.c_int(-1)
    lib.gp_vgicp_batch_get_tuning(batch, _capi.GP_TUNE_EFFECTIVE_MIRROR, C.byref(mirrored))
    in_step = dev_steps.value >= args.steps and dev_stream_us.value > 0 and dev_kernel_us.value > 0
    # `frac` is quoted on the WHOLE kernel the step dispatches -- first workgroup started .. the last part's sums on their way to the host, by the
    # kernel's own 100 MHz stamps over the K timed steps -- not on its streaming slice (kept as frac_streaming)
    streaming_ms = dev_stream_us.value * 1e-3 if in_step else ms_main.value
    kernel_ms = dev_kernel_us.value * 1e-3 if in_step else ms_main.value
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    split = _load_split()

    def _frac(ms):
        return round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms else None

    traffic, traffic_source = _load_traffic()
    traffic_detail = None
    if rank == 0 and world == 1 and not dist_on and not args.no_traffic:
        measured = measure_traffic(args)
        if measured.get("tile_kernel_hbm_bytes_per_launch"):
            traffic, traffic_source, traffic_detail = measured["tile_kernel_hbm_bytes_per_launch"], measured["source"], measured
        else:
            traffic_source = f"{traffic_source}; in-run measurement unavailable ({measured.get('error', 'no counters')})"

    roofline = dict(
        bound="hbm",
        kernel=KERNEL_NAMES.get(_effective_kernel(lib, batch), "?"),
        achieved=round(achieved, 2),
        peak=HBM_PEAK_GBS,
        unit="GB/s",
        frac=round(achieved / HBM_PEAK_GBS, 5),
        frac_note=("frac = algorithmic bytes / the WHOLE fused kernel as the timed steps ran it (first workgroup started .. last part's sums handed to the host; the kernel's own "
                   "100 MHz stamps, mean over the K steps).  Rounds 3-4 quoted the streaming slice under this name: that is frac_streaming now (r04: 0.613 streaming / 0.546 whole kernel)")
        if in_step else "no fused steps in this configuration: frac is the tile kernel back to back under HIP events",
        frac_streaming=_frac(streaming_ms) if in_step else None,
        streaming_ms=round(streaming_ms, 5) if in_step else None,
        streaming_note="first workgroup started .. last partial row in: the part of the kernel the algorithmic bytes belong to (rounds 3-4's `frac`)" if in_step else None,
        traffic=traffic,
        traffic_source=traffic_source,
        traffic_detail=traffic_detail,
        algorithmic_bytes=alg_bytes,
        algorithmic_bytes_note="SURVEY.md 8(d), reference-layout accounting (48 B per source point + the reference's bucket table and voxel arrays): internal repacking does not change it",
        source_stream=("packed private mirror: 36 B per point (12 B point + the 6 floats of the symmetric covariance), three 12-B LDS-DMA rows per 64-point chunk" if mirrored.value == 1
                       else "the caller's arrays: 12 + 36 B per point, four 12-B LDS-DMA rows per chunk"),
        actual_bytes=actual_bytes,
        actual_bytes_note="what the launch requests with perfect reuse of the lookup structures: the source stream as read + 16 B per 4x4x4-voxel block of the map's box + 64-B records + pose and record",
        frac_actual=round(actual_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        frac_fused_kernel=(round(alg_bytes / (dev_kernel_us.value * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if in_step else None),
        kernel_ms=round(kernel_ms, 5),
        kernel_ms_source=(f"measured in THIS run inside the {args.steps} timed steps: the whole fused kernel (first workgroup started .. last part's sums on their way to the host) on the "
                          "device's 100 MHz constant clock, stamped by the kernel itself (gp_vgicp_batch_device_times); mean over the steps") if in_step
        else "measured in THIS run: HIP events over back-to-back tile-kernel launches on the launch stream (no fused steps in this configuration)",
        fused_kernel_ms=round(dev_kernel_us.value * 1e-3, 5) if in_step else None,
        fused_kernel_note="= kernel_ms: the streaming part + the finalize tail of the last eight workgroups" if in_step else None,
        kernel_ms_back_to_back=round(ms_main.value, 5),
        frac_back_to_back=_frac(ms_main.value),
        kernel_ms_back_to_back_source="HIP events on the launch stream over back-to-back launches of the tile kernel (two-kernel form), this run: the kernel at the device's sustained state",
        rocprof_kernel_ms_in_step=split.get("in_step_ms"),
        rocprof_frac_in_step=_frac(split.get("in_step_ms")),
        rocprof_kernel_ms_mean=split.get("all_ms"),
        rocprof_frac_mean=_frac(split.get("all_ms")),
        rocprof_fused_kernel_ms_in_step=split.get("fused_in_step_ms"),
        rocprof_frac_fused_in_step=_frac(split.get("fused_in_step_ms")),
        rocprof_source=split.get("source"),
        step_finalize=args.finalize if not dist_on else "device-resident records (two kernels)",
        finalize_kernel_ms=round(ms_fin.value, 5),
        device_pass_ms=round(ms_total.value, 5),
        cold=cold,
    )

    if os.environ.get("GP_BENCH_CALIBRATE"):  # PMC passes only: known-byte-count stream for FETCH_SIZE calibration
        _capi.check(_capi.load_tune().gp_debug_calibration_stream(src.ptr(src.points_gpu), src.ptr(src.covs_gpu), args.source_points, 5, C.c_void_p(stream.cuda_stream)), "calibration")

    rec = gpa.LinearizedSystem6.from_doubles(host_out[rank].numpy())
    c4 = None
    if not args.no_c4:
        c4 = run_c4(args, lib, gpa, _capi, synthetic, torch, dist, rank, world, device, stream, dist_on)
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        configs = run_configs(args, lib, gpa, _capi, synthetic, torch, device, stream)
    big_source = None
    if rank == 0 and world == 1 and not args.no_configs and not args.no_big_source:
        try:
            big_source = run_big_source(args, lib, gpa, _capi, synthetic, torch, device, stream)
        except Exception as exc:  # the headline must survive this optional leg (memory on a shared box)
            big_source = dict(error=f"{type(exc).__name__}: {exc}")
    result = None
    if rank == 0:
        cpu_baseline = None
        parity = None
        if not args.no_cpu_baseline and world == 1:
            import oracle  # checker / baseline only -- never on the product path

            from oracle import refcapi

            avail = oracle.max_threads()
            use_ref = refcapi.available()  # the reference's own CPU sources (oracle/_ref/libref.so) when they were built
            VM_, VG_ = (refcapi.RefVoxelMap, refcapi.RefVGICPFactor) if use_ref else (oracle.OracleVoxelMap, oracle.OracleVGICPFactor)
            om = VM_(args.resolution)
            om.insert(d["target_points"], d["target_covs"])
            cores = pick_cpu_threads(avail, lambda c: VG_(om, d["source_points"], d["source_covs"], c), lambda o: o.linearize(delta))
            fo = VG_(om, d["source_points"], d["source_covs"], cores)
            Lo = fo.linearize(delta)  # warm-up + parity reference
            parity = {}
            for k in ["H_target", "H_source", "H_target_source", "b_target", "b_source"]:
                parity[k] = float(np.linalg.norm(getattr(rec, k) - getattr(Lo, k)) / np.linalg.norm(getattr(Lo, k)))
            parity["error"] = abs(rec.error - Lo.error) / abs(Lo.error)
            parity["num_inliers_equal"] = bool(rec.num_inliers == Lo.num_inliers)
            times = []
            t_start = time.perf_counter()
            while time.perf_counter() - t_start < args.cpu_seconds * 0.7 or len(times) < 3:
                t = time.perf_counter()
                fo.linearize(delta)
                times.append(time.perf_counter() - t)
            f1 = (refcapi.RefVGICPFactor if use_ref else oracle.OracleVGICPFactor)(om, d["source_points"], d["source_covs"], 1)
            t1 = []
            t_start = time.perf_counter()
            while time.perf_counter() - t_start < args.cpu_seconds * 0.3 or len(t1) < 2:
                t = time.perf_counter()
                f1.linearize(delta)
                t1.append(time.perf_counter() - t)
            med = float(np.median(times))
            cpu_baseline = dict(
                value=round(args.source_points / med, 1),
                unit="point-correspondences/s",
                cores=cores,
                cores_available=avail,
                cores_note="threads chosen by a probe over {all, 1/2, 1/4, 32, 16} of the threads the box reports (pick_cpu_threads): the count that serves the reference's code best",
                kind="reference" if use_ref else "port",
                sample=f"{len(times)} full linearize() passes of the same 1M-pt factor, {cores} OpenMP threads (median {med*1e3:.2f} ms); "
                f"1 thread: {np.median(t1)*1e3:.2f} ms",
                ms_per_linearize=round(med * 1e3, 3),
                ms_per_linearize_1thread=round(float(np.median(t1)) * 1e3, 3),
            )
        result = dict(
            metric="point-correspondences/sec (VGICP linearize, 1M-pt source vs 2M-pt voxel map)",
            value=round(value, 1),
            unit="point-correspondences/s",
            n_gpus=world,
            steps=args.steps,
            warmup=args.warmup,
            ms_per_step=round(ms_per_step, 5),
            ms_per_step_cold=cold["ms_per_step"] if cold else None,
            ms_per_step_cold_note="the same W + K steps measured BEFORE the untimed device wake-up (config.device_warmup), i.e. with --device-warmup-ms 0: rounds 1-3's protocol",
            higher_is_better=True,
            scaling="weak",
            vs_baseline=None,
            dtype="f64",
            dtype_note="transform, fused covariance, its inverse, residual and all reductions in f64; the outer products after the inverse in f32 (kernel family GP_KERNEL_STREAM)",
            data="synthetic",
            config=dict(
                workload="BASELINE configs[1]: single VGICP factor per GPU, 1M synthetic source pts vs 2M-pt GaussianVoxelMap @0.5 m",
                source_points=args.source_points,
                target_points=args.target_points,
                resolution=args.resolution,
                num_voxels=info.num_voxels,
                num_buckets=info.num_buckets,
                inlier_fraction=round(rec.num_inliers / args.source_points, 4),
                parallelism=(f"{world} x 1 factor/GPU; " + ("direct stores of the [N x 122] f64 records into every peer's buffer over xGMI (gp_peer_exchange)" if sharded.exchange == "peer"
                                                             else f"RCCL {sharded.exchange} of stacked [N x 122] f64 records")) if dist_on else "1 GPU",
                exchange=(f"torch.distributed backend {dist.get_backend()}, world {world}, {sharded.exchange}" + (f" (peer exchange not taken: {sharded.peer_note})" if sharded.peer_note else "")
                          if dist_on else None),
                step="poses (host) -> tile kernel -> finalize kernel -> [N>1: ONE RCCL collective over the stacked records: in-place all-gather, or --exchange all_reduce] -> records in host memory, synchronised",
                device_warmup=dict(ms=args.device_warmup_ms, steps=wake_steps,
                                   note="untimed, before the W warm-up steps: the same step run back to back until the device's power state has settled (the first ~10 ms of work "
                                        "behind seconds of host-side set-up run 6-8 % slower: scripts/r04_warm.py, profiles/r04_warm.jsonl), with a torch.cuda.synchronize() every 25 steps -- "
                                        "the timed region ends with one, and a process's first device synchronisations take 50-200 us instead of ~17 (scripts/dbg/k20.py): with K = 20 "
                                        "that alone was +3 us per step"),
            ),
            roofline=roofline,
            cpu_baseline=cpu_baseline,
            parity_vs_oracle=parity,
            c4=c4,
            configs=configs,
            big_source=big_source,
            setup=dict(generate_s=round(t_gen, 2), voxelmap_build_s=round(t_map, 4)),
        )
    lib.gp_vgicp_batch_destroy(batch)
    if dist_on:
        with PhaseGuard(args.phase_seconds, "process group teardown"):
            sharded.close()  # (peer exchange: the peers' buffers are unmapped behind a barrier)
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        if world > 1 and result.get("c4") is not None and not args.no_c4_inlib:
            # VERDICT r04 #5c: the in-library path a C++ optimizer process uses (ONE process drives N devices, no collective: every shard's finalize stores its records into
            # one host-pinned stack) beside the torch.distributed step, per N.  Run when the ranks are gone (the process group is destroyed, their devices idle), in a process
            # of its own with a time limit: neither a hang nor a crash of it may take the line with it
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
        print(json.dumps(result), flush=True)
    return result


if __name__ == "__main__":
    main()
