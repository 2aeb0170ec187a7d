"""bench_detail.py -- the legs of bench.py that are NOT the headline: BASELINE configs[0], [2], [3], [4] (C1, C3, C4, C5 of SURVEY.md 8(d)), the LM loops, the map build and
the 8 M-point source.  bench.py runs them behind the headline while its time budget lasts and writes their objects to bench_detail.json; its last stdout line carries one
number per leg only (the line must stay short enough for the driver to parse).

  c4           -- BASELINE configs[3]: the 4096-factor graph (512 submaps x 32768 points, 8 factors per source submap, 1.0 m voxels) partitioned over the N ranks by source
                  submap with the target maps a shard references replicated onto it; one step = every rank's batched linearise into its rows of the [4096 x 122] f64 stack + ONE
                  collective (RCCL) + D2H.  Strong scaling.
  configs      -- C1 the two full data/kitti_00 scans @0.5 m, C3 the 256-factor submap graph as ONE batched call, C5 k-NN covariance estimation + GICP linearise at 1 M points,
                  the map build; each with ms, corr/s, a roofline object for its dominant kernel, parity against and the time of the REFERENCE's own CPU code
                  (oracle/_ref/libref.so; the C restatement when that is absent).
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BENCH_PY = os.path.join(ROOT, "bench.py")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable)

# The record a job leaves when it dies in an OPTIONAL phase (round 6): once the headline is measured, rank 0 registers the compact line of what it has; a phase that exceeds
# its time box, or a SIGTERM from the launcher tearing the job down because another rank gave up, prints that line as the job's last stdout line before the process ends --
# an untested exchange form that hangs on real hardware must not cost the run its headline.  In a job that ends normally nothing of this is printed.
_LAST_RESORT = {"line": None}


def register_last_resort(line):
    """line: the compact JSON text to print if the job is torn down from here on (None: nothing); installs the SIGTERM hook once"""
    import signal

    first = _LAST_RESORT["line"] is None and "hooked" not in _LAST_RESORT
    _LAST_RESORT["line"] = line
    if first and line is not None:
        _LAST_RESORT["hooked"] = True

        def on_term(_sig, _frm):
            emit_last_resort("SIGTERM")
            os._exit(143)

        try:
            signal.signal(signal.SIGTERM, on_term)
        except Exception:  # (not the main thread: the phase guard still prints)
            pass


def emit_last_resort(why):
    line = _LAST_RESORT.get("line")
    if line:
        _LAST_RESORT["line"] = None
        try:
            obj = json.loads(line)
            obj["aborted"] = str(why)[:96]
            sys.stdout.write(json.dumps(obj, allow_nan=False, separators=(",", ":")) + "\n")
            sys.stdout.flush()
        except Exception:
            pass


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
            emit_last_resort(f"phase '{self.name}' exceeded its time box")
            os._exit(124)

        self._timer = threading.Timer(self.seconds, expire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False


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
        # the exchanged stack against what every rank computed by itself (a private buffer, no exchange), bit for bit, on every rank (collective: objects only)
        verified, bad_rows = None, None
        if dist_on:
            from gtsam_points_amd.distributed import verify_exchanged_stack

            own = torch.zeros((max(n_local, 1), RECORD_DOUBLES), dtype=torch.float64, device=device)
            if n_local:
                issue(poses, own)
            torch.cuda.synchronize()
            verified, bad_rows = verify_exchanged_stack(host_out.numpy(), own.cpu().numpy()[:n_local], begin, end)
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
    if rank == 0 and world == 1 and not dist_on and torch.cuda.device_count() > 1 and not args.no_c4_inlib and getattr(args, "detail", False):  # (minutes: only under --detail)
        # ONE process driving every visible device: in a process of its own with a time limit -- this leg has never run on more than one device (the builder's boxes
        # have one), and neither a hang nor a crash of it may take the headline line with it.  Only in the single-process run (N = 1 on a multi-GPU node): under
        # torch.distributed the other ranks own those devices.
        import subprocess

        try:
            p = subprocess.run([sys.executable, BENCH_PY, "--c4-inlib-only", "--c4-steps", str(args.c4_steps)], capture_output=True, text=True, timeout=300)
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
        exchange=sharded.exchange, exchange_verified=verified, exchange_bad_rows_by_rank=bad_rows, allreduce_ms=round(ar_ms, 4), allgather_ms=round(ag_ms, 4) if ag_ms is not None else None, stack_bytes=F * RECORD_DOUBLES * 8,
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
    """The CPU baseline is the reference's code on THIS box's cores, at the thread count that serves it best: a short probe over {all, 1/2, 32, 16} threads.  The
    container may see more hardware threads than its CPU quota gives it, and a small factor does not scale to hundreds of threads: on one box of round 5 a 22 k-point
    factor took 228 ms with the 256 threads omp_get_max_threads() reported and 1.4 ms with 16.  make(threads) -> object, run(object) = one timed pass."""
    timed = []
    for c in sorted({avail, max(avail // 2, 1), min(avail, 32), min(avail, 16)}):
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


def run_lm_config(workload, gpa, gpu_factors, cpu_factors, pairs, num_poses, truth, values0, sptr, device, cores, kind, cpu_max_iterations=30, solvers=("device", "host", "trial", "native")):
    """configs.lm_*: the reference's LM cadence (bench_lm.py; levenberg_marquardt_ext.cpp:107-143,188-392) over a graph of VGICP factors -- per iteration host to host, by
    phase, on the GPU path (batched linearise, records stay in HBM, block-sparse LL^T on the device; and the same with a host-side numpy solve) and over the checker's CPU
    factors (the reference's own IntegratedVGICPFactor when oracle/_ref is built) as cpu_baseline.  truth None: the CPU run's result is the reference the GPU run is held to.
    cpu_max_iterations < 30: the CPU loop is a BOUNDED sample (its first iterations: the per-iteration cost is what is reported; only with a ground truth to gate against).
    solvers: "device" = damped build + solve as one call (gp_*_system_step), "device-three-calls" = round 4's build / download / solve, "host" = numpy solve."""
    import bench_lm

    cg = bench_lm.CpuGraph(cpu_factors, pairs, num_poses, fixed=0)
    bounded = truth is not None and cpu_max_iterations < 30
    res_cpu = bench_lm.run_lm(cg, values0, max_iterations=cpu_max_iterations if bounded else 30)
    gate_ref = truth if truth is not None else res_cpu["values"]
    cpu = bench_lm.summarize(res_cpu, cg, gate_ref, "cpu")
    out = dict(workload=workload, cadence="linearize(values) -> [solve (A + lambda I) dx = b -> retract -> error(new values) on the linearisation's correspondences] until accepted; "
               "lambda 1e-5, x10 / /10, minModelFidelity 1e-3, relativeErrorTol 1e-5 (GTSAM defaults; levenberg_marquardt_ext.cpp:188-392)",
               gate="max over poses, relative to the fixed pose: rotation < 0.015 rad, translation < 0.15 m (test_matching_cost_factors.cpp:227) against "
               + ("the generator's ground truth" if truth is not None else "the CPU run's result (real scans: no ground truth)"))
    for solver in [x for x in solvers if x not in ("trial", "native")]:
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
        if not bounded:
            obj["pose_vs_cpu_run"] = dict(zip(("rotation_rad", "translation_m"), [round(max(x), 6) for x in zip(*[bench_lm.pose_error(best["values"][k], res_cpu["values"][k]) for k in range(num_poses)])]))
        gg.close()
        out[{"device": "gpu_device_solve", "device-three-calls": "gpu_device_solve_three_calls", "host": "gpu_host_solve"}[solver]] = obj
    if "trial" in solvers or "native" in solvers:
        # round 6: the values in device memory (gp_lm_graph_*): linearise | damped step + retract + error evaluation behind ONE wait
        tg = bench_lm.GpuTrialGraph(gpa, gpu_factors, pairs, num_poses, fixed=0, stream=sptr)
        bench_lm.run_lm(tg, values0, max_iterations=30)
        if "trial" in solvers:
            best = min((bench_lm.run_lm(tg, values0, max_iterations=30) for _ in range(3)), key=lambda r: r["seconds"])
            obj = bench_lm.summarize(best, tg, gate_ref, "gpu, trial on the device (gp_lm_graph_linearize / _try_lambda / _accept driven by the interpreter)")
            tg.sync_phases = True
            split = bench_lm.summarize(bench_lm.run_lm(tg, values0, max_iterations=30), tg, gate_ref, "split")
            tg.sync_phases = False
            obj["ms_per_iteration_by_phase"] = split["ms_per_iteration_by_phase"]
            obj["phase_note"] = "solve = damped step + retract + error evaluation at the trial values, one wait (the error phase is inside it); linearize = its issue + the wait the split run adds"
            out["gpu_device_trial"] = obj
        if "native" in solvers:
            tg.native_loop(values0, max_iterations=30)
            best = min((tg.native_loop(values0, max_iterations=30) for _ in range(5)), key=lambda r: r["seconds"])
            obj = bench_lm.summarize(best, tg, gate_ref, "gpu, the library's own loop (gp_lm_graph_optimize: the reference's cadence over the same three calls, no interpreter inside)")
            obj.pop("ms_per_iteration_by_phase", None)
            obj.pop("dominant_phase", None)
            if not bounded:
                obj["pose_vs_cpu_run"] = dict(zip(("rotation_rad", "translation_m"), [round(max(x), 6) for x in zip(*[bench_lm.pose_error(best["values"][k], res_cpu["values"][k]) for k in range(num_poses)])]))
            out["gpu_native_loop"] = obj
        tg.close()
    cpu.update(cores=cores, kind=kind, sample=(f"the first {cpu_max_iterations} iterations of the loop" if bounded else "the whole loop once") + f": every factor linearised / evaluated in turn with {cores} threads (the count a probe chose, pick_cpu_threads), numpy dense solve")
    out["cpu_baseline"] = cpu
    out["speedup_per_iteration"] = round(cpu["ms_per_iteration"] / out["gpu_native_loop" if "gpu_native_loop" in out else "gpu_device_solve"]["ms_per_iteration"], 1)
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


def _map_build_traffic(points, ms):
    """roofline.traffic of the map build: HBM bytes per build from the committed counter passes (profiles/r06_map_build_pmc.json: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE
    summed over the build's kernels, scripts/r06/map_build_pmc.sh) -- the same cloud and code as here; None when the profile is absent or of another size"""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_map_build_pmc.json")) as f:
            p = json.load(f)
        if p["run"]["points"] != points:
            return dict(traffic=None)
        t = int(p["traffic_bytes_per_build"])
        return dict(traffic=t, traffic_read=int(p["read_bytes"]), traffic_written=int(p["written_bytes"]), frac_traffic=round(t / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    traffic_source="profiles/r06_map_build_pmc.json: rocprofv3 --pmc FETCH_SIZE (x 2, the gfx950 correction) + WRITE_SIZE over every kernel of a build, separate passes; "
                                   "NOT measured in this run (same cloud, same code); 3.7 x the algorithmic bytes: the points are read three times (bounding box, keys, statistics) "
                                   "and the 22-bit sort moves key + index three times")
    except Exception:
        return dict(traffic=None)


def run_configs(args, lib, gpa, _capi, synthetic, torch, device, stream, want=None, target_cloud=None):
    """BASELINE configs[0], [2], [4] (C1, C3, C5 of SURVEY.md 8(d)) under the driver's clock, leg by leg: want(name, estimated_seconds) -> bool decides whether a leg still
    fits the caller's time budget (None = run everything); every object carries the `seconds` its leg took.  GPU side = the product's synchronous entry
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
    if want is None:
        want = lambda name, est: True  # noqa: E731

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
    t_leg = time.time()
    if os.path.exists(os.path.join(gdir, "000000.bin")) and want("C1", 1.5):
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
        out["C1"]["seconds"] = round(time.time() - t_leg, 1)
        t_leg = time.time()
        if not args.no_lm and want("lm_c1", 1.0):
            try:
                out["lm_c1"] = run_lm_config("BASELINE configs[0] as an optimisation: scan 000001 registered to the map of scan 000000 from the identity (one factor, one free pose)",
                                             gpa, [f], [fo], [(0, 1)], 2, None, np.stack([np.eye(4), np.eye(4)]), sptr, device, cores, kind)
            except Exception as exc:  # the headline must survive an optional leg
                out["lm_c1"] = dict(error=f"{type(exc).__name__}: {exc}")
            out["lm_c1"]["seconds"] = round(time.time() - t_leg, 1)
        del f, vm, tgt, src

    # ---- C3: 256-factor submap graph, ONE batched call ----
    def leg_c3():
        t_leg = time.time()
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
        sample = list(range(0, len(factors), 16))  # every 16th factor: 16 reference linearisations (a bounded sample: the CPU side is the checker and a reported baseline)
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
            for _ in range(2):  # (the better of two behind a warm-up pass: the first pass behind other host work pays for waking the team)
                tt = time.perf_counter()
                Lo = fo.linearize(g["deltas"][k])
                reps.append(time.perf_counter() - tt)
            t_cpu += float(np.min(reps))
            par = _parity(gpa.LinearizedSystem6.from_doubles(recs[k]), Lo)
            worst = max(worst, max(par[b] for b in BLOCKS), par["error"])
            assert par["num_inliers_equal"], k
        cpu_ms_graph = t_cpu / len(sample) * len(factors) * 1e3
        out["C3"] = dict(
            workload="BASELINE configs[2]: 256-factor submap graph (64 submaps x ~22k pts, factors i -> i+1..i+4 and back, 1.0 m voxels), ONE batched linearise "
                     "through gp_vgicp_batch_linearize_view",
            factors=len(factors), points=npts, ms=round(ms_view, 5), ms_with_copy=round(ms_copy, 5), corr_per_s=round(npts / ms_view * 1e3, 1), roofline=roof,
            cpu_baseline=dict(value=round(npts / cpu_ms_graph * 1e3, 1), unit="point-correspondences/s", cores=cores, cores_available=avail, kind=kind, ms=round(cpu_ms_graph, 2),
                              sample=f"{len(sample)} of the 256 factors (every 16th), the better of two linearize() passes each after a warm-up, {cores} threads per factor, sequential over factors "
                                     "as graph_.linearize does; scaled x16"),
            parity_vs_reference_max=worst, parity_factors_checked=len(sample), inlier_fraction=round(float(recs[:, 0].sum()) / npts, 4), setup_s=round(t_setup, 1))
        t_lm = time.time()
        if not args.no_lm and want("lm_c3", 5.0):
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
                                             gpa, factors, cpu_factors, g["pairs"], n_sub, truth, v0, sptr, device, cores, kind,
                                             cpu_max_iterations=30 if getattr(args, "lm_full_cpu", False) else 2,
                                             solvers=("device", "device-three-calls", "host", "trial", "native") if getattr(args, "lm_full_cpu", False) else ("device", "host", "trial", "native"))
                del cpu_factors
            except Exception as exc:
                out["lm_c3"] = dict(error=f"{type(exc).__name__}: {exc}")
            out["lm_c3"]["seconds"] = round(time.time() - t_lm, 1)
        del factors, maps, clouds
        return round(time.time() - t_leg, 1)

    # ---- C5: k-NN covariance estimation (k = 10) + IntegratedGICPFactor linearise, 1 M points ----
    def leg_c5():
        t_leg = time.time()
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
        # a fraction for the covariance call.  Bound = VALU issue: the call's three launches execute a fixed number of vector-ALU wave-instructions on
        # this cloud (rocprofv3 --pmc SQ_INSTS_VALU per launch, profiles/r06_c5_pmc.json: counted once, the code and the cloud are the same here), every one of which
        # holds a SIMD's issue port for four clocks; peak = 1024 SIMDs x 2.4 GHz / 4.  achieved = those instructions / THIS run's wall per call (structure build included)
        try:
            with open(os.path.join(ROOT, "profiles", "r06_c5_pmc.json")) as f_pmc:
                pmc = json.load(f_pmc)["launches"]
            per_call = sum((2.0 if name.startswith("covariance_kernel") else 1.0) * (v["counters"]["SQ_INSTS_VALU"] + 3.0 * v["counters"].get("SQ_INSTS_VALU_TRANS_F64", 0.0)) for name, v in pmc.items())
            peak = 1024 * 2.4e9 / 4.0
            cov_roof.update(bound="valu-issue", unit="G wave-instructions/s", achieved=round(per_call / (cov_ms * 1e-3) / 1e9, 2), peak=round(peak / 1e9, 1),
                            frac=round(per_call / (cov_ms * 1e-3) / peak, 4), valu_wave_instructions_per_call=int(per_call),
                            frac_by_launch={name: v.get("frac_valu_issue") for name, v in pmc.items()},
                            frac_source="instructions per call from profiles/r06_c5_pmc.json (rocprofv3 --pmc SQ_INSTS_VALU, transcendentals x4; the heavy and the light "
                                        "launch of covariance_kernel counted once each) / this run's wall per call; frac_by_launch: each launch against its own rocprofv3 duration")
        except Exception as exc:  # (the profile is evidence, not a dependency)
            cov_roof.update(frac=None, frac_source=f"profiles/r06_c5_pmc.json not readable: {exc}")
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
        return round(time.time() - t_leg, 1)

    # ---- map build: the Gaussian voxel map of the 2 M-point C2 target at 0.5 m (replaces types/gaussian_voxelmap_gpu.cu:211-307), wall per gp_voxelmap_insert ----
    def leg_map_build():
        t_leg = time.time()
        tgt2 = target_cloud  # (the headline's own 2 M-point target cloud, already resident, when bench.py hands it over)
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
                          algorithmic_bytes=48 * nt, **_map_build_traffic(nt, mb_ms),
                          note="48 B per point read once (SURVEY.md 8(d), voxel-map build); host wall of the whole call (10 launches, three points where the host waits), not one kernel: the build is "
                               "a chain of latency-bound kernels at this size (DESIGN.md section 4.4)"))
        return round(time.time() - t_leg, 1)

    for name, est, leg in (("C3", 7.0, leg_c3), ("C5", 6.0, leg_c5), ("map_build", 2.5, leg_map_build)):
        if want(name, est):
            try:
                secs = leg()
                out[name]["seconds"] = secs
            except Exception as exc:  # the headline must survive an optional leg
                out[name] = dict(error=f"{type(exc).__name__}: {exc}")
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
