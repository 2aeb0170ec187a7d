"""Shared helpers of the test-suite (parity metrics, tiny LM driver)."""
import numpy as np

BLOCKS = ["H_target", "H_source", "H_target_source", "b_target", "b_source"]


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))


def assert_linearized_close(got, ref, tol, what=""):
    """norm-wise relative parity on every block (||dH||_F/||H||_F, ||db||_2/||b||_2), error and inlier count"""
    for k in BLOCKS:
        e = rel_err(getattr(got, k), getattr(ref, k) if not isinstance(ref, dict) else ref[k])
        assert e <= tol, f"{what} {k}: relative error {e:.3e} > {tol:.1e}"
    ref_err = ref["error"] if isinstance(ref, dict) else ref.error
    ref_inl = ref["num_inliers"] if isinstance(ref, dict) else ref.num_inliers
    assert abs(got.error - ref_err) <= tol * max(abs(ref_err), 1e-300), f"{what} error {got.error} vs {ref_err}"
    assert got.num_inliers == ref_inl, f"{what} num_inliers {got.num_inliers} vs {ref_inl}"


def expmap(xi):
    from gtsam_points_amd.synthetic import expmap as e

    return e(xi)


def pose_error(T_est, T_gt):
    d = np.linalg.inv(T_gt) @ T_est
    ang = np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))
    return float(ang), float(np.linalg.norm(d[:3, 3]))


def lm_optimize(linearize_all, error_all, values, keys, fixed=(), max_iter=30, rel_tol=1e-4, lam0=1e-5):
    """Minimal Levenberg-Marquardt over SE(3) poses with right perturbations (GTSAM Pose3 retract = Expmap here),
    mirroring LevenbergMarquardtOptimizerExt's loop shape (levenberg_marquardt_ext.cpp:354-398): linearise via the
    hook, try lambdas until the nonlinear error decreases.
      linearize_all(values) -> list of HessianFactor-like (keys, G blocks, g, f)
      error_all(values)     -> total nonlinear error  (sum of factor errors, GTSAM's 0.5 factor is irrelevant here)"""
    free = [k for k in keys if k not in fixed]
    idx = {k: i for i, k in enumerate(free)}
    lam = lam0
    values = dict(values)
    factors = linearize_all(values)
    err = sum(f.f for f in factors)
    for _ in range(max_iter):
        n = 6 * len(free)
        H = np.zeros((n, n))
        g = np.zeros(n)
        for f in factors:
            for (i, j), blk in f.G.items():
                ki, kj = f.keys[i], f.keys[j]
                if ki in idx and kj in idx:
                    a, b = idx[ki] * 6, idx[kj] * 6
                    H[a : a + 6, b : b + 6] += blk
                    if i != j:
                        H[b : b + 6, a : a + 6] += blk.T
            for i, gi in enumerate(f.g):
                if f.keys[i] in idx:
                    a = idx[f.keys[i]] * 6
                    g[a : a + 6] += gi
        improved = False
        for _try in range(12):
            dx = np.linalg.solve(H + lam * np.diag(np.diag(H)) + 1e-9 * np.eye(n), g)
            new_values = dict(values)
            for k in free:
                new_values[k] = values[k] @ expmap(dx[idx[k] * 6 : idx[k] * 6 + 6])
            # evaluate with correspondences frozen at the linearisation point (error() semantics)
            new_err = error_all(new_values)
            if new_err < err:
                improved = True
                lam = max(lam / 10.0, 1e-12)
                break
            lam *= 10.0
        if not improved:
            break
        rel = (err - new_err) / max(err, 1e-300)
        values = new_values
        factors = linearize_all(values)
        err = sum(f.f for f in factors)
        if rel < rel_tol:
            break
    return values
