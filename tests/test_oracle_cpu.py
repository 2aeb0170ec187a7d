"""CPU tests of the oracle itself (test infrastructure): golden vectors, C vs numpy, analytic vs numeric derivatives,
and the reference's own alignment gate (src/test/test_matching_cost_factors.cpp:196-230)."""
import os

import numpy as np
import pytest

import oracle
from oracle import vgicp_oracle_np as onp
from helpers import BLOCKS, assert_linearized_close, expmap, lm_optimize, pose_error, rel_err

REF_DATA = "/root/reference/data"


def _factor(d, res, threads=1):
    vm = oracle.OracleVoxelMap(res)
    vm.insert(d["target_points"], d["target_covs"])
    return vm, oracle.OracleVGICPFactor(vm, d["source_points"], d["source_covs"], threads)


@pytest.mark.parametrize("name,res", [("kitti00_dec8_r0.5_identity", 0.5), ("kitti00_dec8_r0.5_c1b", 0.5), ("kitti00_dec8_r1.0_c1b", 1.0)])
def test_oracle_matches_golden(kitti00, golden, name, res):
    g = golden[name]
    vm, f = _factor(kitti00, res)
    assert vm.num_voxels == g["num_voxels"]
    L = f.linearize(np.array(g["delta"]))
    assert_linearized_close(L, g, 1e-12, name)
    if "delta_eval" in g:
        e = f.error(np.array(g["delta_eval"]))
        assert abs(e - g["error_eval"]) <= 1e-12 * abs(g["error_eval"])


def test_oracle_threads_agree(kitti00, golden):
    g = golden["kitti00_dec8_r0.5_c1b"]
    _, f1 = _factor(kitti00, 0.5, 1)
    _, f4 = _factor(kitti00, 0.5, 4)
    L1, L4 = f1.linearize(np.array(g["delta"])), f4.linearize(np.array(g["delta"]))
    assert_linearized_close(L4, L1, 1e-12, "threads")


def test_c_vs_numpy_restatement(kitti00):
    delta = expmap([0.02, 0.01, -0.015, -0.08, 0.06, 0.02])
    vm, f = _factor(kitti00, 0.5)
    L = f.linearize(delta)
    vn = onp.VoxelMapNP(0.5)
    vn.insert(kitti00["target_points"], kitti00["target_covs"].transpose(0, 2, 1).reshape(-1, 9))
    Ln = onp.vgicp_linearize(vn, kitti00["source_points"], kitti00["source_covs"].transpose(0, 2, 1).reshape(-1, 9), delta)
    assert vm.num_voxels == vn.num_voxels
    for k in BLOCKS:
        assert rel_err(getattr(L, k), Ln[k]) < 1e-10
    assert L.num_inliers == Ln["num_inliers"]
    assert abs(L.error - Ln["error"]) < 1e-10 * abs(Ln["error"])


def test_voxelmap_statistics(kitti00):
    """voxel = mean of points / mean of covariances; first-seen order (gaussian_voxelmap_cpu.cpp:23-47)"""
    vm = oracle.OracleVoxelMap(0.5)
    vm.insert(kitti00["target_points"], kitti00["target_covs"])
    coords, num_points, means, covs, _ = vm.export()
    p = kitti00["target_points"].astype(np.float64)
    c = onp.fast_floor(p * 2.0)
    assert num_points.sum() == len(p)
    for v in [0, 1, len(coords) // 2, len(coords) - 1]:
        sel = (c == coords[v]).all(1)
        assert sel.sum() == num_points[v]
        np.testing.assert_allclose(means[v], p[sel].mean(0), rtol=0, atol=1e-12)
        np.testing.assert_allclose(covs[v], kitti00["target_covs"][sel].astype(np.float64).mean(0), rtol=0, atol=1e-12)


def test_fast_floor_negative_and_exact():
    x = np.array([-1.5, -1.0, -0.0, 0.0, 0.999999, 1.0, -2.000001, 2.5])
    np.testing.assert_array_equal(onp.fast_floor(x), np.floor(x).astype(np.int64))


def test_b_is_half_gradient_and_H_is_gauss_newton(kitti00):
    """b_s = 1/2 d/dxi_s sum r^T M r with correspondences and M frozen (error() semantics); H symmetric PSD"""
    delta = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    _, f = _factor(kitti00, 1.0)
    L = f.linearize(delta)
    eps = 1e-6
    for block, side in [("b_source", "s"), ("b_target", "t")]:
        g = np.zeros(6)
        for k in range(6):
            xi = np.zeros(6)
            xi[k] = eps
            if side == "s":
                dp, dm = delta @ expmap(xi), delta @ expmap(-xi)
            else:  # target perturbation: (T_t Exp(xi))^-1 T_s = Exp(-xi) delta
                dp, dm = expmap(-xi) @ delta, expmap(xi) @ delta
            g[k] = (f.error(dp) - f.error(dm)) / (2 * eps)
        assert rel_err(0.5 * g, getattr(L, block)) < 1e-5
    for H in [L.H_target, L.H_source]:
        assert rel_err(H, H.T) < 1e-12
        assert np.linalg.eigvalsh(0.5 * (H + H.T)).min() > -1e-6 * np.abs(H).max()


def test_adjoint_identity(kitti00):
    """J_s = -J_t Ad(delta)  =>  H_s = Ad^T H_t Ad, H_ts = -H_t Ad, b_s = -Ad^T b_t  (used by the HIP finalize kernel)"""
    delta = expmap([0.03, -0.02, 0.05, 0.4, -0.3, 0.1])
    _, f = _factor(kitti00, 0.5)
    L = f.linearize(delta)
    R, t = delta[:3, :3], delta[:3, 3]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ad = np.block([[R, np.zeros((3, 3))], [tx @ R, R]])
    assert rel_err(Ad.T @ L.H_target @ Ad, L.H_source) < 1e-11
    assert rel_err(-L.H_target @ Ad, L.H_target_source) < 1e-11
    assert rel_err(-Ad.T @ L.b_target, L.b_source) < 1e-11


def test_knn_matches_bruteforce():
    """test_kdtree.cpp:92-163: 1000 uniform points in [-100,100]^3, k in {1,2,3,5,10,15,20}, sq-dists within 1e-6"""
    rng = np.random.default_rng(0)
    p = rng.uniform(-100, 100, (1000, 3)).astype(np.float32)
    q = rng.uniform(-100, 100, (200, 3)).astype(np.float32)
    tree = oracle.OracleKdTree(p)
    d2 = ((q[:, None, :].astype(np.float64) - p[None].astype(np.float64)) ** 2).sum(2)
    for k in [1, 2, 3, 5, 10, 15, 20]:
        idx, d = tree.knn(q, k)
        ref = np.sort(d2, 1)[:, :k]
        assert np.abs(d - ref).max() < 1e-6
        assert (np.take_along_axis(d2, idx, 1) - d).max() < 1e-9
    idx, d = tree.knn(q, 5, max_sq_dist=100.0)
    assert ((d < 100.0) == (np.sort(d2, 1)[:, :5] < 100.0)).all()


def test_covariances_c_vs_numpy(kitti00):
    p = kitti00["source_points"][:4000]
    c, short = oracle.estimate_covariances(p, 10, 2)
    cn, _ = onp.estimate_covariances(p, 10)
    assert short == 0
    d = np.abs(c - cn).reshape(len(p), -1).max(1)
    assert np.percentile(d, 99) < 1e-7  # degenerate neighbourhoods (eigenvector choice) are the tail
    w = np.linalg.eigvalsh(0.5 * (c + c.transpose(0, 2, 1)))
    np.testing.assert_allclose(np.median(w, 0), [1e-3, 1.0, 1.0], atol=1e-9)


def test_eig3_direct_against_eigh():
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.normal(size=(3, 3))
        m = a @ a.T
        ev, V = oracle.capi.eig3_direct(m)
        np.testing.assert_allclose(ev, np.linalg.eigvalsh(m), rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(V @ np.diag(ev) @ V.T, m, atol=1e-8)


def test_alignment_gate_oracle(kitti07):
    """the reference's end-to-end gate: LM from poses perturbed by Expmap(U(-0.1,0.1)^6) must come back within
    0.015 rad / 0.15 m of ground truth (test_matching_cost_factors.cpp:104-108,196-230), 5-frame chain, voxel 1.0 m"""
    from gtsam_points_amd.factors import HessianFactor

    poses = kitti07["poses"]
    rng = np.random.default_rng(8191)
    vms, facs = [], []
    for i in range(5):
        vm = oracle.OracleVoxelMap(1.0)
        vm.insert(kitti07[f"points_{i}"], kitti07[f"covs_{i}"])
        vms.append(vm)
    pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 2), (1, 3), (2, 4)]
    for i, j in pairs:
        facs.append(oracle.OracleVGICPFactor(vms[i], kitti07[f"points_{j}"], kitti07[f"covs_{j}"], 2))
    values = {k: poses[k] @ expmap(rng.uniform(-0.1, 0.1, 6)) if k > 0 else poses[0].copy() for k in range(5)}

    def lin(vals):
        out = []
        for (i, j), f in zip(pairs, facs):
            L = f.linearize(oracle.calc_delta(vals[i], vals[j]))
            out.append(HessianFactor([i, j], {(0, 0): L.H_target, (0, 1): L.H_target_source, (1, 1): L.H_source}, [-L.b_target, -L.b_source], L.error))
        return out

    def err(vals):
        return sum(f.error(oracle.calc_delta(vals[i], vals[j])) for (i, j), f in zip(pairs, facs))

    est = lm_optimize(lin, err, values, list(range(5)), fixed=(0,))
    for k in range(1, 5):
        ang, trans = pose_error(np.linalg.inv(est[0]) @ est[k], np.linalg.inv(poses[0]) @ poses[k])
        assert ang < 0.015 and trans < 0.15, (k, ang, trans)


@pytest.mark.skipif(not os.path.exists(REF_DATA), reason="reference data not mounted (GPU box)")
def test_oracle_on_full_c1_anchor(golden):
    """full kitti_00 pair, 10,970 voxels @0.5 m, overlap 0.773 at identity"""
    tgt = np.fromfile(f"{REF_DATA}/kitti_00/000000.bin", dtype=np.float32).reshape(-1, 3)
    src = np.fromfile(f"{REF_DATA}/kitti_00/000001.bin", dtype=np.float32).reshape(-1, 3)
    ct, _ = oracle.estimate_covariances(tgt, 10, 8)
    cs, _ = oracle.estimate_covariances(src, 10, 8)
    sym = lambda c: np.ascontiguousarray(0.5 * (c.astype(np.float32) + c.astype(np.float32).transpose(0, 2, 1)))
    vm = oracle.OracleVoxelMap(0.5)
    vm.insert(tgt, sym(ct))
    assert vm.num_voxels == 10970
    assert abs(vm.overlap(src) - 0.7732) < 1e-3
    f = oracle.OracleVGICPFactor(vm, src, sym(cs), 8)
    for name in ["kitti00_full_r0.5_identity", "kitti00_full_r0.5_c1b"]:
        g = golden[name]
        assert_linearized_close(f.linearize(np.array(g["delta"])), g, 1e-11, name)
