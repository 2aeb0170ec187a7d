"""GPU parity tests at the sizes BASELINE.json quotes: every configuration of BASELINE.json.configs is
run through the C-ABI here and checked against the reference-pinned oracle -- every factor, not a sample.

  C1  the two full data/kitti_00 scans (124,668 / 124,605 points; shipped as tests/golden/kitti_00/*.bin), 0.5 m voxels,
      identity and the C1(b) perturbation: committed golden vectors (tests/golden/make_fixtures.py) + the oracle
  C2  lives in test_vgicp_gpu.py::test_linearity_and_determinism_at_1m (1 M vs 2 M points) and in bench.py
  C3  256-factor submap graph, ONE batched linearise: all 256 factors vs the oracle
  C4  the whole 4096-factor configuration (512 submaps x 32768 points): one plain batch == the in-library 8-shard plan bit for
      bit, all 4096 records and error evaluations vs the oracle
  C5  k-NN covariances of 1 M points + the GICP linearise of 1 M vs 1 M points
  +   covariances that are NOT symmetric (lower triangle 1-2 ulp off the upper), the case the symmetrised fixtures never exercise

Tolerance: PARITY_TOL = 1e-6 on every block of every factor (gate: 1e-5)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from helpers import BLOCKS, assert_linearized_close, expmap, rel_err

pytestmark = pytest.mark.gpu
PARITY_TOL = 1e-6
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sym32(c):
    c = c.astype(np.float32)
    return np.ascontiguousarray(0.5 * (c + c.transpose(0, 2, 1)))


def _sync_linearize(gpu, factor, delta):
    rec = gpu._capi.Linearized6()
    gpu._capi.check(factor._lib.gp_vgicp_factor_linearize(factor._h, gpu.types._pose16(delta), C.byref(rec)), "linearize")
    return gpu.LinearizedSystem6(rec)


def _batch_linearize(gpu, factors, deltas):
    lib = gpu.load()
    F = len(factors)
    arr = (C.c_void_p * F)(*[f._h.value for f in factors])
    batch, s = C.c_void_p(), C.c_void_p()
    gpu._capi.check(lib.gp_stream_create(C.byref(s)), "stream")
    gpu._capi.check(lib.gp_vgicp_batch_create(arr, F, s, C.byref(batch)), "batch")
    poses = np.stack([np.ascontiguousarray(np.asarray(d).T).reshape(16) for d in deltas]).copy()
    out = np.zeros((F, 122))
    gpu._capi.check(lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out.ctypes.data), "batch_linearize")
    out2 = np.zeros((F, 122))
    gpu._capi.check(lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out2.ctypes.data), "batch_linearize")
    assert np.array_equal(out, out2)  # fixed summation order: bit-reproducible
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)
    return [gpu.LinearizedSystem6.from_doubles(out[k]) for k in range(F)]


# ---- C1 ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def kitti00_full():
    tp = np.fromfile(os.path.join(GOLDEN, "kitti_00", "000000.bin"), dtype=np.float32).reshape(-1, 3)
    sp = np.fromfile(os.path.join(GOLDEN, "kitti_00", "000001.bin"), dtype=np.float32).reshape(-1, 3)
    assert len(tp) == 124668 and len(sp) == 124605
    tc, short_t = oracle.estimate_covariances(tp, 10, oracle.max_threads())
    sc, short_s = oracle.estimate_covariances(sp, 10, oracle.max_threads())
    assert short_t == 0 and short_s == 0
    return dict(target_points=tp, target_covs=_sym32(tc), source_points=sp, source_covs=_sym32(sc))  # as make_fixtures.py


@pytest.mark.parametrize("name", ["kitti00_full_r0.5_identity", "kitti00_full_r0.5_c1b"])
def test_c1_full_kitti00_scans(gpu, kitti00_full, golden, name):
    g = golden[name]
    d = kitti00_full
    tgt = gpu.PointCloudGPU(d["target_points"], d["target_covs"])
    src = gpu.PointCloudGPU(d["source_points"], d["source_covs"])
    vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    assert vm.voxelmap_info.num_voxels == g["num_voxels"] == 10970  # 10,970 voxels at 0.5 m
    f = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src)
    delta = np.array(g["delta"])
    L = _sync_linearize(gpu, f, delta)
    assert_linearized_close(L, g, PARITY_TOL, name + " vs golden")
    om = oracle.OracleVoxelMap(0.5)
    om.insert(d["target_points"], d["target_covs"])
    fo = oracle.OracleVGICPFactor(om, d["source_points"], d["source_covs"], oracle.max_threads())
    assert_linearized_close(L, fo.linearize(delta), PARITY_TOL, name + " vs oracle")
    if "delta_eval" in g:
        out = C.c_double()
        gpu._capi.check(f._lib.gp_vgicp_factor_compute_error(f._h, gpu.types._pose16(delta), gpu.types._pose16(np.array(g["delta_eval"])), C.byref(out)), "error")
        assert abs(out.value - g["error_eval"]) <= PARITY_TOL * abs(g["error_eval"])
    # the k-NN covariances of the GPU path on the same scans (C1 states "estimate_covariances(k=10)"): same factor, covariances
    # straight from gp_estimate_covariances without any host-side symmetrisation
    tgt2, src2 = gpu.PointCloudGPU(d["target_points"]), gpu.PointCloudGPU(d["source_points"])
    assert gpu.estimate_covariances_gpu(tgt2, 10) == 0 and gpu.estimate_covariances_gpu(src2, 10) == 0
    vm2 = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm2.insert(tgt2)
    L2 = _sync_linearize(gpu, gpu.IntegratedVGICPFactorGPU(0, 1, vm2, src2), delta)
    tc2, sc2 = tgt2.download("covs"), src2.download("covs")  # float32 exactly as the kernels read them
    om2 = oracle.OracleVoxelMap(0.5)
    om2.insert(d["target_points"], tc2)
    assert_linearized_close(L2, oracle.OracleVGICPFactor(om2, d["source_points"], sc2, oracle.max_threads()).linearize(delta), PARITY_TOL, name + " GPU covariances")
    # eigenvector choice differs in degenerate neighbourhoods, so L2 is only close to L, not equal
    assert L2.num_inliers == L.num_inliers and rel_err(L2.H_source, L.H_source) < 0.05


# ---- C3 ---------------------------------------------------------------------------------------------------------------------
def _check_all_factors(gpu, clouds_host, pairs, deltas, res, label):
    clouds = {i: gpu.PointCloudGPU(p, c) for i, (p, c) in clouds_host.items()}
    maps, omaps = {}, {}
    for t in sorted({t for t, _ in pairs}):
        m = gpu.GaussianVoxelMapGPU(res, target_points_drop_rate=0.0)
        m.insert(clouds[t])
        maps[t] = m
        om = oracle.OracleVoxelMap(res)
        om.insert(*clouds_host[t])
        omaps[t] = om
        assert m.voxelmap_info.num_voxels == om.num_voxels
    factors = [gpu.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s]) for t, s in pairs]
    Ls = _batch_linearize(gpu, factors, deltas)
    worst, inl = 0.0, 0
    for k, ((t, s), delta, L) in enumerate(zip(pairs, deltas, Ls)):
        Lo = oracle.OracleVGICPFactor(omaps[t], clouds_host[s][0], clouds_host[s][1], oracle.max_threads()).linearize(delta)
        assert L.num_inliers == Lo.num_inliers, (label, k)
        inl += L.num_inliers
        if Lo.num_inliers == 0:
            continue
        for blk in BLOCKS:
            worst = max(worst, rel_err(getattr(L, blk), getattr(Lo, blk)))
        assert abs(L.error - Lo.error) <= PARITY_TOL * abs(Lo.error), (label, k)
    assert worst <= PARITY_TOL, f"{label}: worst relative error over {len(pairs)} factors {worst:.3e}"
    return worst, inl


def test_c3_256_factor_graph_every_factor(gpu):
    from gtsam_points_amd import synthetic

    g = synthetic.make_c3_graph()
    assert len(g["pairs"]) == 256
    host = {i: pc for i, pc in enumerate(g["clouds"])}
    worst, inl = _check_all_factors(gpu, host, g["pairs"], g["deltas"], 1.0, "C3")
    total = sum(len(host[s][0]) for _, s in g["pairs"])
    assert inl > 0.5 * total  # the graph really overlaps


def test_c4_all_4096_factors_plain_and_sharded(gpu):
    """BASELINE configs[3], the WHOLE configuration: 4096 factors (512 submaps x 32768 points, 8 outgoing factors per source
    submap, 1.0 m voxels).  Every factor is linearised twice on the GPU -- through one plain batch of 4096 and through the
    in-library sharded path (gp_vgicp_multi_batch_* with the 8-shard gp_shard_plan a node of eight GPUs would use; on a 1-GPU box
    the eight shards share the device, on a multi-GPU node they spread) -- the two must agree bit for bit, and every one of the 4096
    records and error evaluations is held against the oracle (the loop being replaced:
    src/gtsam_points/cuda/nonlinear_factor_set_gpu.cpp:64-139)."""
    from gtsam_points_amd import synthetic
    from gtsam_points_amd.distributed import MultiDeviceBatch, partition_factors

    pairs = synthetic.c4_factor_pairs()
    F = len(pairs)
    assert F == 4096
    sub = synthetic.make_c4_submaps(range(synthetic.C4_SUBMAPS))
    clouds = {i: gpu.PointCloudGPU(sub[i][0], sub[i][1]) for i in range(synthetic.C4_SUBMAPS)}
    maps = {}
    for t in sorted({t for t, _ in pairs}):
        m = gpu.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        m.insert(clouds[t])
        maps[t] = m
    factors = [gpu.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s]) for t, s in pairs]
    deltas = [synthetic.c4_delta(sub, t, s) for t, s in pairs]
    rng = np.random.default_rng(77)
    deltas_e = [d @ expmap(rng.uniform(-0.005, 0.005, 6)) for d in deltas]

    # plain batch: one table of 4096 factors
    lib = gpu.load()
    arr = (C.c_void_p * F)(*[f._h.value for f in factors])
    batch, s = C.c_void_p(), C.c_void_p()
    gpu._capi.check(lib.gp_stream_create(C.byref(s)), "stream")
    gpu._capi.check(lib.gp_vgicp_batch_create(arr, F, s, C.byref(batch)), "batch")
    poses = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in deltas]).copy()
    poses_e = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in deltas_e]).copy()
    plain, plain_err = np.zeros((F, 122)), np.zeros(F)
    gpu._capi.check(lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, plain.ctypes.data), "batch_linearize")
    gpu._capi.check(lib.gp_vgicp_batch_compute_error(batch, poses.ctypes.data, poses_e.ctypes.data, plain_err.ctypes.data), "batch_error")
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)

    # the in-library sharded path with the plan of an 8-GPU node
    parts = partition_factors([synthetic.C4_POINTS] * F, 8)
    assert [e - b for b, e in parts] == [512] * 8  # equal weights: eight shards of 64 source submaps each
    shard_of = np.zeros(F, np.int32)
    for k, (b, e) in enumerate(parts):
        shard_of[b:e] = k
    mb = MultiDeviceBatch(factors, shard_of=shard_of, num_shards=8, use_rccl=0)
    assert mb.num_shards == 8 and all(mb.shard_info(k)["num_factors"] == 512 for k in range(8))
    sharded = mb.linearize(deltas)
    sharded_err = mb.compute_error(deltas, deltas_e)
    assert np.array_equal(sharded, plain), "sharded != unsharded on the whole C4 configuration"
    assert np.array_equal(sharded_err, plain_err)
    del mb

    # every factor against the oracle
    omaps = {}
    worst, worst_err, inl = 0.0, 0.0, 0
    per_shard_inl = np.zeros(8)
    for k, ((t, sidx), delta) in enumerate(zip(pairs, deltas)):
        if t not in omaps:
            om = oracle.OracleVoxelMap(1.0)
            om.insert(sub[t][0], sub[t][1])
            assert maps[t].voxelmap_info.num_voxels == om.num_voxels
            omaps[t] = om
        fo = oracle.OracleVGICPFactor(omaps[t], sub[sidx][0], sub[sidx][1], oracle.max_threads())
        Lo = fo.linearize(delta)
        L = gpu.LinearizedSystem6.from_doubles(plain[k])
        assert L.num_inliers == Lo.num_inliers, k
        inl += L.num_inliers
        per_shard_inl[k // 512] += L.num_inliers
        if Lo.num_inliers == 0:
            continue
        for blk in BLOCKS:
            worst = max(worst, rel_err(getattr(L, blk), getattr(Lo, blk)))
        assert abs(L.error - Lo.error) <= PARITY_TOL * abs(Lo.error), k
        eo = fo.error(deltas_e[k])
        worst_err = max(worst_err, abs(plain_err[k] - eo) / abs(eo))
    assert worst <= PARITY_TOL, f"C4: worst relative error over 4096 factors {worst:.3e}"
    assert worst_err <= PARITY_TOL, f"C4: worst error-evaluation mismatch over 4096 factors {worst_err:.3e}"
    frac = per_shard_inl / (512 * synthetic.C4_POINTS)
    assert inl > 0.4 * F * synthetic.C4_POINTS and frac.min() > 0.3, frac  # every shard really overlaps


# ---- C5 ---------------------------------------------------------------------------------------------------------------------
def test_c5_knn_covariances_and_gicp_at_1m(gpu):
    from gtsam_points_amd import synthetic

    d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
    tgt, src = gpu.PointCloudGPU(d["target_points"]), gpu.PointCloudGPU(d["source_points"])
    assert gpu.estimate_covariances_gpu(tgt, 10) == 0 and gpu.estimate_covariances_gpu(src, 10) == 0
    got = src.download("covs").astype(np.float64)
    ref, short = oracle.estimate_covariances(d["source_points"], 10, oracle.max_threads())
    assert short == 0
    rel = np.linalg.norm((got - ref).reshape(len(got), -1), axis=1) / np.linalg.norm(ref.reshape(len(got), -1), axis=1)
    assert np.median(rel) < 2e-7 and (rel < 1e-5).mean() > 0.995  # <= 1e-5 except flagged degenerate neighbourhoods
    # GICP: 1-NN correspondences within 1.0 m and the same algebra, 1 M vs 1 M points, covariances as the GPU produced them
    tc, sc = tgt.download("covs"), src.download("covs")
    f = gpu.IntegratedGICPFactorGPU(0, 1, tgt, src)
    fo = oracle.OracleGICPFactor(d["target_points"], tc, d["source_points"], sc, oracle.max_threads())
    delta = d["T_true"] @ expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    L, Lo = f.linearize_delta(delta), fo.linearize(delta)
    assert_linearized_close(L, Lo, PARITY_TOL, "C5 GICP 1M")
    assert L.num_inliers > 0.5 * len(d["source_points"])
    # a pose whose 3x3 block is not orthonormal (6-digit quaternions upstream) takes the 92-sum path and must match as well
    skew = delta.copy()
    skew[:3, :3] = skew[:3, :3] @ (np.eye(3) + 1e-6 * np.array([[1.0, 0.3, 0.0], [0.0, -0.5, 0.2], [0.1, 0.0, 0.7]]))
    assert_linearized_close(f.linearize_delta(skew), fo.linearize(skew), PARITY_TOL, "C5 GICP 1M, general pose")


# ---- covariances that are not symmetric ---------------------------------------------------------------------------------------
def test_unsymmetrised_covariances(gpu, kitti00):
    """every committed fixture is symmetrised on the host; the kernels read all nine entries and use the symmetric part of a
    non-symmetric matrix (gtsam_points_hip.h, Conventions).  Here the lower triangle of every third covariance is moved 1-2 ulp
    off the upper one (what a float cast of V diag V^-1 can produce).  The reference CPU factor uses the full 3x3
    (integrated_vgicp_factor_impl.hpp:138-140) and a HessianFactor keeps the upper triangle of H
    (integrated_matching_cost_factor.cpp:49): compare the symmetric parts of H, and b / error as they are."""
    rng = np.random.default_rng(5)

    def perturb(c):
        c = c.copy()
        pick = np.arange(len(c)) % 3 == 0
        for (a, b) in [(1, 0), (2, 0), (2, 1)]:
            steps = rng.integers(1, 3, size=pick.sum()) * rng.choice([-1, 1], size=pick.sum())
            v = c[pick, a, b].view(np.int32) + steps.astype(np.int32)
            c[pick, a, b] = v.view(np.float32)
        return c

    tc, sc = perturb(kitti00["target_covs"]), perturb(kitti00["source_covs"])
    assert (tc != tc.transpose(0, 2, 1)).any() and (sc != sc.transpose(0, 2, 1)).any()
    tgt = gpu.PointCloudGPU(kitti00["target_points"], tc)
    src = gpu.PointCloudGPU(kitti00["source_points"], sc)
    np.testing.assert_array_equal(src.download("covs"), sc)  # all nine entries reach the device as given
    delta = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    om = oracle.OracleVoxelMap(0.5)
    om.insert(kitti00["target_points"], tc)
    Lo = oracle.OracleVGICPFactor(om, kitti00["source_points"], sc, 2).linearize(delta)
    for variant in [0, 2, 8, 12]:  # GP_KERNEL_*: reference-shaped kernel, hashed pipeline, the round-2 grid kernel, stream (default)
        vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
        vm.insert(tgt)
        L = _sync_linearize(gpu, gpu.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(0, variant), delta)
        assert L.num_inliers == Lo.num_inliers
        for k in ["H_target", "H_source"]:
            h = getattr(Lo, k)
            assert rel_err(getattr(L, k), 0.5 * (h + h.T)) <= PARITY_TOL, (variant, k)
        for k in ["H_target_source", "b_target", "b_source"]:
            assert rel_err(getattr(L, k), getattr(Lo, k)) <= PARITY_TOL, (variant, k)
        assert abs(L.error - Lo.error) <= PARITY_TOL * Lo.error
    # GICP reads both clouds' covariances the same way
    fg = gpu.IntegratedGICPFactorGPU(0, 1, tgt, src)
    Lg = fg.linearize_delta(delta)
    Lgo = oracle.OracleGICPFactor(kitti00["target_points"], tc, kitti00["source_points"], sc, 2).linearize(delta)
    assert Lg.num_inliers == Lgo.num_inliers
    for k in ["H_target", "H_source"]:
        h = getattr(Lgo, k)
        assert rel_err(getattr(Lg, k), 0.5 * (h + h.T)) <= PARITY_TOL
    assert rel_err(Lg.b_source, Lgo.b_source) <= PARITY_TOL
