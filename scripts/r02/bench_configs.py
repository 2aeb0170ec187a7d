"""Measures the BASELINE.json parity-test configurations that are not the bench.py headline:
  C1  two kitti_00-style scans (fixture kitti00_dec8), 0.5 m voxels, single linearise
  C3  256-factor submap graph (64 submaps x ~25k pts, 1.0 m voxels, factors i -> i+1..i+4), ONE batched linearise
  C4  one GPU's shard of the 4096-factor / 8-GPU configuration: 512 factors x 32768 pts, 1.0 m voxels
For each: ms per linearise (synchronous product call), corr/s, algorithmic roofline fraction of the tile kernel
(HIP events), parity vs the oracle (all factors for C1/C3, a sample for C4) and the oracle's own time.
Output: one JSON object per config on stdout (copied to profiles/rNN_configs.jsonl)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gtsam_points_amd as gpa  # noqa: E402
import oracle  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

lib = gpa.load()
if os.environ.get("GP_VARIANT"):  # tuning: pick a tile-kernel variant (gp_debug_set_variant)
    _capi.check(lib.gp_debug_set_variant(int(os.environ["GP_VARIANT"])), "variant")
if os.environ.get("GP_TILE_INTERLEAVE"):  # tuning: execution order of the tiles of factors that share a source cloud
    _capi.check(lib.gp_debug_set_tile_interleave(int(os.environ["GP_TILE_INTERLEAVE"])), "interleave")
if os.environ.get("GP_XCD_CHUNK"):  # tuning: workgroup -> tile map (gp_debug_set_xcd_chunk)
    _capi.check(lib.gp_debug_set_xcd_chunk(int(os.environ["GP_XCD_CHUNK"])), "xcd chunk")
VARIANTS = [int(v) for v in os.environ["GP_VARIANTS"].split(",")] if os.environ.get("GP_VARIANTS") else [None]  # A/B of tile-kernel variants in one process
ONLY = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else {"C1", "C3", "C4", "C5"}
BLOCKS = ["H_target", "H_source", "H_target_source", "b_target", "b_source"]


def run(name, clouds, maps, pairs, deltas, host_clouds, res, oracle_sample, iters=20):
    for v in VARIANTS:
        if v is not None:
            _capi.check(lib.gp_debug_set_variant(v), "variant")
        run_variant(name if v is None else f"{name} [variant {v}]", clouds, maps, pairs, deltas, host_clouds, res, oracle_sample, iters)


def run_variant(name, clouds, maps, pairs, deltas, host_clouds, res, oracle_sample, iters=20):
    factors = [gpa.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j]) for i, j in pairs]
    F = len(factors)
    arr = (C.c_void_p * F)(*[f._h.value for f in factors])
    batch, s = C.c_void_p(), C.c_void_p()
    lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, F, s, C.byref(batch)), "batch")
    poses = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in deltas]).copy()
    out = np.zeros((F, 122))
    for _ in range(3):
        _capi.check(lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out.ctypes.data), "lin")
    def wall(call):  # per-call host-to-host times: the median is the figure, mean and max show a hiccup of the box when there is one
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            call()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts)), float(np.mean(ts)), float(np.max(ts))

    view = C.c_void_p()
    ms, ms_mean, ms_max = wall(lambda: lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out.ctypes.data))
    ms_view, ms_view_mean, ms_view_max = wall(lambda: lib.gp_vgicp_batch_linearize_view(batch, poses.ctypes.data, C.byref(view)))
    got = np.ctypeslib.as_array(C.cast(view, C.POINTER(C.c_double)), shape=(F, 122))
    assert np.array_equal(got, out)  # the view holds the same records the copying call delivered
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    _capi.check(lib.gp_vgicp_batch_time_linearize(batch, poses.ctypes.data, iters, C.byref(a), C.byref(b), C.byref(c)), "time")
    npts = int(lib.gp_vgicp_batch_total_points(batch))
    alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
    if os.environ.get("GP_TRACE_TILES"):  # tuning: per-workgroup s_memtime phases of the rolling-DMA kernel (variants 1, 2)
        trace = torch.zeros((65536, 8), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        _capi.check(lib.gp_debug_set_trace_buffer(C.c_void_p(trace.data_ptr())), "trace")
        lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out.ctypes.data)
        torch.cuda.synchronize()
        lib.gp_debug_set_trace_buffer(None)
        t = trace.cpu().numpy().astype(np.float64)
        t = t[(t[:, 0] > 0) & (t[:, 7] > 0)]
        dur = np.diff(t, axis=1)
        print("traced", len(t), "lifetime median ticks", np.median(t[:, 7] - t[:, 0]), "phases median", [int(np.median(dur[:, k])) for k in range(7)],
              "p90", [int(np.percentile(dur[:, k], 90)) for k in range(7)], flush=True)
    worst, t_cpu = 0.0, 0.0
    omaps = {}
    for k in oracle_sample:
        i, j = pairs[k]
        if i not in omaps:
            om = oracle.OracleVoxelMap(res)
            om.insert(*host_clouds[i])
            omaps[i] = om
        fo = oracle.OracleVGICPFactor(omaps[i], host_clouds[j][0], host_clouds[j][1], oracle.max_threads())
        t = time.perf_counter()
        Lo = fo.linearize(deltas[k])
        t_cpu += time.perf_counter() - t
        L = gpa.LinearizedSystem6.from_doubles(out[k])
        assert L.num_inliers == Lo.num_inliers, (name, k)
        for blk in BLOCKS:
            worst = max(worst, float(np.linalg.norm(getattr(L, blk) - getattr(Lo, blk)) / np.linalg.norm(getattr(Lo, blk))))
    res_d = dict(
        config=name, factors=F, points=npts, ms_per_linearize=round(ms, 4), ms_per_linearize_view=round(ms_view, 4), ms_per_linearize_mean_max=[round(ms_mean, 4), round(ms_max, 4)],
        ms_per_linearize_view_mean_max=[round(ms_view_mean, 4), round(ms_view_max, 4)], corr_per_s=round(npts / ms * 1e3, 1), tile_kernel_ms=round(b.value, 5),
        finalize_kernel_ms=round(c.value, 5), device_pass_ms=round(a.value, 5), algorithmic_bytes=alg, roofline_frac=round(alg / (b.value * 1e-3) / 8e12, 4),
        parity_max_rel_err=worst, parity_factors_checked=len(oracle_sample),
        cpu_oracle_ms_per_factor=round(t_cpu / max(len(oracle_sample), 1) * 1e3, 3), cpu_threads=oracle.max_threads(),
        inlier_fraction=round(float(out[:, 0].sum()) / npts, 4),
    )
    print(json.dumps(res_d), flush=True)
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)


# ---- C1 ----
if "C1" in ONLY:
    k = np.load(os.path.join(ROOT, "tests/golden/kitti00_dec8.npz"))
    tgt = gpa.PointCloudGPU(k["target_points"], k["target_covs"])
    src = gpa.PointCloudGPU(k["source_points"], k["source_covs"])
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    run("C1 kitti00 (every 8th point), 0.5 m, single linearise", [tgt, src], [vm, None], [(0, 1)], [synthetic.expmap(synthetic.C1B_PERTURBATION)],
        [(k["target_points"], k["target_covs"]), (k["source_points"], k["source_covs"])], 0.5, [0], iters=200)

# ---- C3 ----
if "C3" in ONLY or "C4" in ONLY:
    rng = np.random.default_rng(8191)
    walls, stations = synthetic.make_street(64, spacing=6.0, seed=43)
    host, clouds, maps = [], [], []
    t0 = time.time()
    for i, T in enumerate(stations):
        p, c, _ = synthetic.make_submap(20000 + 80 * i, seed=1000 + i, walls=walls, sensor_pose=T)
        host.append((p, c))
        clouds.append(gpa.PointCloudGPU(p, c))
        m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        m.insert(clouds[-1])
        maps.append(m)
    pairs = [(i, j) for i in range(64) for j in range(i + 1, min(i + 5, 64))]
    pairs = (pairs + [(j, i) for i, j in pairs])[:256]
    deltas = [np.linalg.inv(stations[i]) @ stations[j] @ synthetic.expmap(rng.uniform(-0.02, 0.02, 6)) for i, j in pairs]
    run(f"C3 256-factor submap graph, 64 submaps x ~22k pts, 1.0 m (setup {time.time()-t0:.1f}s)", clouds, maps, pairs, deltas, host, 1.0, list(range(0, 256, 16)))

# ---- C4 shard (one GPU of eight): 512 factors x 32768 points ----
if "C4" in ONLY:
    t0 = time.time()
    base_host, base_clouds = [], []
    for i in range(64):
        p, c, _ = synthetic.make_submap(32768, seed=2000 + i, walls=walls, sensor_pose=stations[i])
        base_host.append((p, c))
    host4, clouds4, maps4 = [], [], []
    g = torch.Generator(device="cuda").manual_seed(44)
    for i in range(512):
        p, c = base_host[i % 64]
        if i < 64:
            ph = p
        else:  # distinct memory and slightly different geometry per replica (deterministic jitter), so that nothing is shared in cache
            ph = (p.astype(np.float64) + np.random.default_rng(3000 + i).normal(0, 0.01, p.shape)).astype(np.float32)
        host4.append((ph, c))
        clouds4.append(gpa.PointCloudGPU(ph, c))
        m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        m.insert(clouds4[-1])
        maps4.append(m)
    pairs4 = [(i, (i // 64) * 64 + (i % 64 + 1) % 64) for i in range(512)]
    deltas4 = [np.linalg.inv(stations[i % 64]) @ stations[j % 64] @ synthetic.expmap(rng.uniform(-0.02, 0.02, 6)) for i, j in pairs4]
    run(f"C4 shard: 512 factors x 32768 pts (1/8 of the 4096-factor config), 1.0 m (setup {time.time()-t0:.1f}s)", clouds4, maps4, pairs4, deltas4, host4, 1.0, list(range(0, 512, 64)))

# ---- C5: k-NN covariance estimation + GICP linearise, 1 M points ----
if "C5" in ONLY:
    d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
    tgt5 = gpa.PointCloudGPU(d["target_points"])
    src5 = gpa.PointCloudGPU(d["source_points"])
    torch.cuda.synchronize()
    t_cov = []
    for fr in (tgt5, src5, tgt5, src5):
        t = time.perf_counter()
        short = gpa.estimate_covariances_gpu(fr, 10)
        t_cov.append(time.perf_counter() - t)
    t_cov_gpu = min(t_cov[2:])
    t = time.perf_counter()
    oc_src, _ = oracle.estimate_covariances(d["source_points"], 10, oracle.max_threads())
    t_cov_cpu = time.perf_counter() - t
    got = src5.covs_gpu.cpu().numpy().reshape(-1, 3, 3).transpose(0, 2, 1).astype(np.float64)
    rel = np.linalg.norm((got - oc_src).reshape(len(got), -1), axis=1) / np.linalg.norm(oc_src.reshape(len(got), -1), axis=1)
    fg = gpa.IntegratedGICPFactorGPU(0, 1, tgt5, src5)
    delta5 = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    L = fg.linearize_delta(delta5)
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        L = fg.linearize_delta(delta5)
        ts.append(time.perf_counter() - t)
    tc = tgt5.covs_gpu.cpu().numpy().reshape(-1, 3, 3).transpose(0, 2, 1)
    sc = src5.covs_gpu.cpu().numpy().reshape(-1, 3, 3).transpose(0, 2, 1)
    fo = oracle.OracleGICPFactor(d["target_points"], tc, d["source_points"], sc, oracle.max_threads())
    t = time.perf_counter()
    Lo = fo.linearize(delta5)
    t_gicp_cpu = time.perf_counter() - t
    worst = max(float(np.linalg.norm(getattr(L, b) - getattr(Lo, b)) / np.linalg.norm(getattr(Lo, b))) for b in BLOCKS)
    print(json.dumps(dict(
        config="C5 kNN covariance estimation (k=10) + IntegratedGICPFactor linearise, 1M pts vs 1M pts", points=1_000_000,
        cov_estimation_ms=round(t_cov_gpu * 1e3, 3), cov_points_per_s=round(1e6 / t_cov_gpu, 1), cov_cpu_oracle_ms=round(t_cov_cpu * 1e3, 1),
        cov_rel_err_median=float(np.median(rel)), cov_frac_within_1e5=float((rel < 1e-5).mean()), cov_num_short=short,
        gicp_linearize_ms=round(float(np.median(ts)) * 1e3, 4), gicp_corr_per_s=round(1e6 / float(np.median(ts)), 1), gicp_cpu_oracle_ms=round(t_gicp_cpu * 1e3, 1),
        gicp_parity_max_rel_err=worst, gicp_inliers_equal=bool(L.num_inliers == Lo.num_inliers), gicp_inlier_fraction=round(L.num_inliers / 1e6, 4), cpu_threads=oracle.max_threads(),
    )), flush=True)
