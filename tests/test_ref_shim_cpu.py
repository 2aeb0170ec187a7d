"""The pin under the pin.  oracle/_ref/libref.so is the reference's own control flow compiled on top of STAND-IN Eigen / GTSAM headers
(oracle/ref_shim/include, written for this repository because neither library is on the image): every Matrix::inverse(), SelfAdjointEigenSolver, Pose3::inverse /
compose / matrix, SO3::Hat, Isometry3d::inverse inside it is the stand-in's arithmetic.  Here that arithmetic is held, on its own, against an independent implementation
(numpy / LAPACK) on 10^4 random inputs each, to 1e-13 of the result's scale -- including the cases that matter: covariances of condition 10^3 (the regularised (1e-3, 1, 1)
ones the factor inverts), near-degenerate eigenvalue pairs, large translations.  Call sites stood in for:
include/gtsam_points/factors/impl/integrated_vgicp_factor_impl.hpp:138-140,233-237; src/gtsam_points/features/covariance_estimation.cpp:49-53;
src/gtsam_points/factors/integrated_matching_cost_factor.cpp:59-66."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libshimtest.so")
N = 10_000


@pytest.fixture(scope="module")
def shim():
    assert os.path.exists(LIB), "build it: make -C oracle (python -c 'import __graft_entry__ as g; g.build()')"
    return C.CDLL(LIB)


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _rot(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(n, 3, 3)


def _spd(rng, n, evals):
    R = _rot(rng, n)
    return R @ (evals[:, :, None] * R.transpose(0, 2, 1))


def test_inverse3_and_the_fused_covariance(shim):
    """Matrix3d::inverse() (cofactor closed form) on: the fused covariances the factor inverts -- C_B + R C_A R^T with both operands of eigenvalues (1e-3, 1, 1) up to
    generic SPD ones (condition <= 1e4) -- and general well-conditioned matrices"""
    rng = np.random.default_rng(1)
    ev = np.stack([10.0 ** rng.uniform(-3.5, 0.5, N), 10.0 ** rng.uniform(-0.5, 0.5, N), 10.0 ** rng.uniform(-0.5, 0.5, N)], axis=1)
    ev[: N // 2] = [1e-3, 1.0, 1.0]
    CA, CB, R = _spd(rng, N, ev), _spd(rng, N, ev[::-1].copy()), _rot(rng, N)
    S = np.zeros((N, 3, 3))
    shim.shim_sandwich3(_ptr(np.ascontiguousarray(R)), _ptr(np.ascontiguousarray(CA)), _ptr(np.ascontiguousarray(CB)), N, _ptr(S))
    S_np = CB + R @ CA @ R.transpose(0, 2, 1)
    assert np.abs(S - S_np).max() <= 1e-14 * np.abs(S_np).max()
    for A in (S_np, rng.normal(size=(N, 3, 3)) + 3.0 * np.eye(3)):
        A = np.ascontiguousarray(A)
        out = np.zeros_like(A)
        shim.shim_inverse3(_ptr(A), N, _ptr(out))
        ref = np.linalg.inv(A)
        # residual form (independent of either inverse's conditioning): ||A X - I|| <= cond * eps, and entry-wise against LAPACK
        resid = np.abs(A @ out - np.eye(3)).reshape(N, -1).max(axis=1)
        cond = np.linalg.cond(A)
        assert (resid <= 2e-15 * cond).all(), float((resid / cond).max())
        scale = np.abs(ref).reshape(N, -1).max(axis=1)
        assert (np.abs(out - ref).reshape(N, -1).max(axis=1) <= 1e-13 * scale * np.maximum(cond / 100.0, 1.0)).all()


def test_inverse4_and_pose_algebra(shim):
    """Pose3(A).inverse() * Pose3(B) (integrated_matching_cost_factor.cpp:59-66), .matrix(), Isometry3d::inverse(), Matrix4d::inverse() against numpy on rigid poses with
    translations up to 1 km (the large-coordinate case the f64 path exists for)"""
    rng = np.random.default_rng(2)
    def poses():
        T = np.tile(np.eye(4), (N, 1, 1))
        T[:, :3, :3] = _rot(rng, N)
        T[:, :3, 3] = rng.uniform(-1.0, 1.0, (N, 3)) * 10.0 ** rng.uniform(-1, 3, (N, 1))
        return np.ascontiguousarray(T)
    A, B = poses(), poses()
    d, ia, di = np.zeros_like(A), np.zeros_like(A), np.zeros_like(A)
    shim.shim_pose_ops(_ptr(A), _ptr(B), N, _ptr(d), _ptr(ia), _ptr(di))
    ref_ia = np.linalg.inv(A)
    ref_d = ref_ia @ B
    tscale = np.maximum(np.abs(A[:, :3, 3]).max(axis=1) + np.abs(B[:, :3, 3]).max(axis=1), 1.0)[:, None, None]
    assert (np.abs(ia - ref_ia) <= 1e-13 * tscale).all()
    assert (np.abs(d - ref_d) <= 1e-13 * tscale).all() and (np.abs(di - ref_d) <= 1e-13 * tscale).all()
    assert np.array_equal(d[:, 3], np.tile([0.0, 0.0, 0.0, 1.0], (N, 1)))
    G = np.ascontiguousarray(rng.normal(size=(N, 4, 4)) + 4.0 * np.eye(4))
    out = np.zeros_like(G)
    shim.shim_inverse4(_ptr(G), N, _ptr(out))
    assert (np.abs(G @ out - np.eye(4)).reshape(N, -1).max(axis=1) <= 1e-14 * np.linalg.cond(G)).all()


def test_hat_and_expmap(shim):
    rng = np.random.default_rng(3)
    xi = rng.uniform(-1.0, 1.0, (N, 6)) * 10.0 ** rng.uniform(-9, 0.4, (N, 1))  # rotation angles from 1e-9 (the series branch) to ~2.5 rad
    hat, E = np.zeros((N, 3, 3)), np.zeros((N, 4, 4))
    shim.shim_hat_expmap(_ptr(np.ascontiguousarray(xi)), N, _ptr(hat), _ptr(E))
    w = xi[:, :3]
    ref_hat = np.zeros((N, 3, 3))
    ref_hat[:, 0, 1], ref_hat[:, 0, 2], ref_hat[:, 1, 0], ref_hat[:, 1, 2], ref_hat[:, 2, 0], ref_hat[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    assert np.array_equal(hat, ref_hat)
    from scipy.linalg import expm

    pick = rng.choice(N, 400, replace=False)
    for i in pick:
        X = np.zeros((4, 4))
        X[:3, :3] = ref_hat[i]
        X[:3, 3] = xi[i, 3:]
        assert np.abs(E[i] - expm(X)).max() <= 2e-14 * max(1.0, np.abs(xi[i, 3:]).max()), i
    R = E[:, :3, :3]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() <= 1e-14 and np.abs(np.linalg.det(R) - 1.0).max() <= 1e-14


def test_self_adjoint_eigensolver_including_near_degenerate_pairs(shim):
    """SelfAdjointEigenSolver<Matrix3d>::computeDirect's stand-in (a cyclic Jacobi iteration) against numpy.linalg.eigh: eigenvalues to 1e-13 of the largest, the
    decomposition V diag(l) V^T = A and V^T V = I to 1e-13, eigenvectors themselves wherever the gap allows -- on generic spectra, on k-NN sample covariances' typical
    (flat: one small eigenvalue) spectra, and on pairs of eigenvalues that coincide to 1e-3 .. 1e-14 relative (where the eigenVECTORS of the pair are arbitrary, the
    invariant subspace and what covariance_estimation.cpp:49-53 builds from it are not)"""
    rng = np.random.default_rng(4)
    ev = np.sort(10.0 ** rng.uniform(-4, 1, (N, 3)), axis=1)
    ev[N // 3: 2 * N // 3, 0] *= 1e-3                                  # flat neighbourhoods
    gap = 10.0 ** rng.uniform(-14, -3, N - 2 * N // 3)
    ev[2 * N // 3:, 2] = ev[2 * N // 3:, 1] * (1.0 + gap)  # near-degenerate upper pair
    ev = np.sort(ev, axis=1)
    A = _spd(rng, N, ev)
    A = np.ascontiguousarray(0.5 * (A + A.transpose(0, 2, 1)))
    evals, V = np.zeros((N, 3)), np.zeros((N, 3, 3))
    shim.shim_eig3(_ptr(A), N, _ptr(evals), _ptr(V))
    w_np, V_np = np.linalg.eigh(A)
    scale = np.abs(w_np).max(axis=1, keepdims=True)
    assert (np.diff(evals, axis=1) >= 0).all()
    assert (np.abs(evals - w_np) <= 1e-13 * scale).all()
    assert np.abs(V.transpose(0, 2, 1) @ V - np.eye(3)).max() <= 1e-13
    recon = V @ (evals[:, :, None] * V.transpose(0, 2, 1))
    assert (np.abs(recon - A).reshape(N, -1).max(axis=1) <= 1e-13 * scale[:, 0]).all()
    # eigenvectors up to sign where every gap is >= 1e-6 of the scale
    gaps = np.minimum(np.diff(w_np, axis=1).min(axis=1), np.inf) / scale[:, 0]
    ok = gaps >= 1e-6
    dots = np.abs(np.einsum("nrk,nrk->nk", V[ok], V_np[ok]))
    assert ok.sum() > N // 2 and (1.0 - dots <= 1e-10 / gaps[ok][:, None] * 1e-3 + 1e-12).all()
    # what the covariance estimation makes of it: V diag(1e-3, 1, 1) V^T depends only on the eigenvector of the SMALLEST eigenvalue -- well defined whenever the lower
    # gap is, whatever the upper pair does
    low_gap = (w_np[:, 1] - w_np[:, 0]) / scale[:, 0]
    sel = low_gap >= 1e-6
    lam = np.array([1e-3, 1.0, 1.0])
    C_shim = V[sel] @ (lam[None, :, None] * V[sel].transpose(0, 2, 1))
    C_np = V_np[sel] @ (lam[None, :, None] * V_np[sel].transpose(0, 2, 1))
    assert sel.sum() > 0.9 * N and (np.abs(C_shim - C_np).reshape(sel.sum(), -1).max(axis=1) <= 1e-9 / low_gap[sel] * 1e-6 + 1e-12).all()
