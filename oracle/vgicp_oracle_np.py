"""Independent numpy restatement of the reference's CPU VGICP linearisation.

TEST INFRASTRUCTURE ONLY.  Second opinion for oracle/vgicp_oracle.c: written from the
formulas (not from the C file), vectorised over points, 3-vector / 3x3 algebra instead of
the reference's homogeneous 4-vectors.  Follows (paths relative to /root/reference):
  include/gtsam_points/factors/impl/integrated_vgicp_factor_impl.hpp:99-257
  src/gtsam_points/types/gaussian_voxelmap_cpu.cpp:23-77
  include/gtsam_points/util/fast_floor.hpp:12-15
"""
import numpy as np


def fast_floor(x):
    """int(x) - (x < int(x)), fast_floor.hpp:13-14 (truncate toward zero, then fix negatives)."""
    n = np.trunc(x).astype(np.int64)
    return (n - (x < n)).astype(np.int64)


def hat(v):
    """SO3::Hat for an (N,3) batch -> (N,3,3)."""
    v = np.atleast_2d(v)
    z = np.zeros(len(v))
    return np.stack(
        [np.stack([z, -v[:, 2], v[:, 1]], -1), np.stack([v[:, 2], z, -v[:, 0]], -1), np.stack([-v[:, 1], v[:, 0], z], -1)], -2
    )


def expmap(xi):
    """GTSAM Pose3::Expmap, xi = [omega, v]."""
    xi = np.asarray(xi, dtype=np.float64)
    w, v = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = hat(w)[0]
    if th < 1e-10:
        R = np.eye(3) + W
        V = np.eye(3)
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th**2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


class VoxelMapNP:
    """GaussianVoxelMapCPU: dict keyed by voxel coordinate, voxels in first-seen order."""

    def __init__(self, resolution):
        self.resolution = float(resolution)
        self.inv_leaf = 1.0 / float(resolution)

    def insert(self, points, covs):
        p = np.asarray(points, dtype=np.float32).astype(np.float64)
        c = np.asarray(covs, dtype=np.float32).astype(np.float64).reshape(-1, 3, 3)
        coords = fast_floor(p * self.inv_leaf)
        uniq, first_idx, inv = np.unique(coords, axis=0, return_index=True, return_inverse=True)
        inv = inv.reshape(-1)
        order = np.argsort(first_idx, kind="stable")  # first-seen order, like flat_voxels
        rank = np.empty_like(order)
        rank[order] = np.arange(len(order))
        vid = rank[inv]
        V = len(uniq)
        self.coords = uniq[order]
        self.num_points = np.bincount(vid, minlength=V)
        self.means = np.zeros((V, 3))
        self.covs = np.zeros((V, 3, 3))
        np.add.at(self.means, vid, p)
        np.add.at(self.covs, vid, c)
        self.means /= self.num_points[:, None]
        self.covs /= self.num_points[:, None, None]
        self.index = {tuple(k): i for i, k in enumerate(self.coords.tolist())}

    @property
    def num_voxels(self):
        return len(self.coords)

    def lookup(self, q):
        coords = fast_floor(q * self.inv_leaf)
        return np.array([self.index.get(tuple(k), -1) for k in coords.tolist()], dtype=np.int64)


def vgicp_linearize(vmap, points, covs, delta, delta_eval=None):
    """linearize() = update_correspondences(delta) + evaluate(delta_eval or delta).

    Returns dict(num_inliers, error, H_target, H_source, H_target_source, b_target, b_source).
    """
    delta = np.asarray(delta, dtype=np.float64)
    de = delta if delta_eval is None else np.asarray(delta_eval, dtype=np.float64)
    p = np.asarray(points, dtype=np.float32).astype(np.float64)
    CA = np.asarray(covs, dtype=np.float32).astype(np.float64).reshape(-1, 3, 3)
    R, t = delta[:3, :3], delta[:3, 3]
    q_l = p @ R.T + t
    vid = vmap.lookup(q_l)
    ok = vid >= 0
    p, CA, vid = p[ok], CA[ok], vid[ok]
    muB, CB = vmap.means[vid], vmap.covs[vid]
    RCR = CB + R @ CA @ R.T
    M = np.linalg.inv(RCR)
    Re, te = de[:3, :3], de[:3, 3]
    q = p @ Re.T + te
    r = muB - q
    Mr = np.einsum("nij,nj->ni", M, r)
    err = float(np.einsum("ni,ni->", r, Mr))
    Jt = np.concatenate([-hat(q), np.broadcast_to(np.eye(3), (len(q), 3, 3))], axis=2)
    Js = np.concatenate([Re @ hat(p), np.broadcast_to(-Re, (len(q), 3, 3))], axis=2)
    JtM = np.einsum("nki,nkj->nij", Jt, M)
    JsM = np.einsum("nki,nkj->nij", Js, M)
    return dict(
        num_inliers=int(ok.sum()),
        error=err,
        H_target=np.einsum("nik,nkj->ij", JtM, Jt),
        H_source=np.einsum("nik,nkj->ij", JsM, Js),
        H_target_source=np.einsum("nik,nkj->ij", JtM, Js),
        b_target=np.einsum("nik,nk->i", JtM, r),
        b_source=np.einsum("nik,nk->i", JsM, r),
    )


def estimate_covariances(points, k=10):
    """covariance_estimation.cpp:18-77 via scipy cKDTree + numpy eigh (independent of the C kd-tree
    and of Eigen's direct solver): I - 0.999 n n^T form is NOT assumed; V diag(1e-3,1,1) V^T is used."""
    from scipy.spatial import cKDTree

    p = np.asarray(points, dtype=np.float32).astype(np.float64)
    tree = cKDTree(p)
    _, idx = tree.query(p, k=k)
    nb = p[idx]  # (N,k,3)
    s = nb.sum(1)
    ss = np.einsum("nki,nkj->nij", nb, nb)
    mean = s / k
    cov = (ss - mean[:, :, None] * s[:, None, :]) / k
    cov = 0.5 * (cov + cov.transpose(0, 2, 1))
    w, V = np.linalg.eigh(cov)
    lam = np.array([1e-3, 1.0, 1.0])
    return np.einsum("nik,k,njk->nij", V, lam, V), idx
