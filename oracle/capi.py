"""ctypes bindings of oracle/liboracle.so (the C restatement, vgicp_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile (gcc -O3 -fopenmp)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "vgicp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class _Lin6(C.Structure):
    _fields_ = [
        ("num_inliers", C.c_int),
        ("pad_", C.c_int),
        ("error", C.c_double),
        ("H_target", C.c_double * 36),
        ("H_source", C.c_double * 36),
        ("H_target_source", C.c_double * 36),
        ("b_target", C.c_double * 6),
        ("b_source", C.c_double * 6),
    ]


class Linearized6:
    """Result record: 6x6 blocks as numpy (row, col) arrays, double."""

    def __init__(self, num_inliers, error, H_target, H_source, H_target_source, b_target, b_source):
        self.num_inliers = int(num_inliers)
        self.error = float(error)
        self.H_target = np.asarray(H_target, dtype=np.float64)
        self.H_source = np.asarray(H_source, dtype=np.float64)
        self.H_target_source = np.asarray(H_target_source, dtype=np.float64)
        self.b_target = np.asarray(b_target, dtype=np.float64)
        self.b_source = np.asarray(b_source, dtype=np.float64)

    @staticmethod
    def from_struct(s):
        def m(a):
            return np.frombuffer(a, dtype=np.float64).reshape(6, 6).T.copy()  # column-major -> (r, c)

        return Linearized6(
            s.num_inliers,
            s.error,
            m(s.H_target),
            m(s.H_source),
            m(s.H_target_source),
            np.frombuffer(s.b_target, dtype=np.float64).copy(),
            np.frombuffer(s.b_source, dtype=np.float64).copy(),
        )


def _lib():
    global _LIB
    if _LIB is None:
        so = build()
        lib = C.CDLL(so)
        vp, dp, fp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)
        lib.orc_calc_delta.argtypes = [dp, dp, dp]
        lib.orc_pose_expmap.argtypes = [dp, dp]
        lib.orc_voxelmap_create.restype = vp
        lib.orc_voxelmap_create.argtypes = [C.c_double]
        lib.orc_voxelmap_destroy.argtypes = [vp]
        lib.orc_voxelmap_insert.argtypes = [vp, fp, fp, fp, C.c_int]
        lib.orc_voxelmap_num_voxels.argtypes = [vp]
        lib.orc_voxelmap_num_voxels.restype = C.c_int
        lib.orc_voxelmap_lookup.argtypes = [vp, dp]
        lib.orc_voxelmap_lookup.restype = C.c_int
        lib.orc_voxelmap_lookup_coord.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        lib.orc_voxelmap_lookup_coord.restype = C.c_int
        lib.orc_voxelmap_export.argtypes = [vp, ip, ip, dp, dp, dp]
        lib.orc_voxelmap_overlap.argtypes = [vp, fp, C.c_int, dp]
        lib.orc_voxelmap_overlap.restype = C.c_double
        lib.orc_vgicp_create.restype = vp
        lib.orc_vgicp_create.argtypes = [vp, fp, fp, C.c_int, C.c_int]
        lib.orc_vgicp_destroy.argtypes = [vp]
        lib.orc_vgicp_update_correspondences.argtypes = [vp, dp]
        lib.orc_vgicp_evaluate.argtypes = [vp, dp, C.POINTER(_Lin6)]
        lib.orc_vgicp_evaluate.restype = C.c_double
        lib.orc_vgicp_linearize.argtypes = [vp, dp, C.POINTER(_Lin6)]
        lib.orc_vgicp_error.argtypes = [vp, dp]
        lib.orc_vgicp_error.restype = C.c_double
        lib.orc_vgicp_num_inliers.argtypes = [vp]
        lib.orc_vgicp_num_inliers.restype = C.c_int
        lib.orc_vgicp_correspondences.argtypes = [vp]
        lib.orc_vgicp_correspondences.restype = ip
        lib.orc_kdtree_create.restype = vp
        lib.orc_kdtree_create.argtypes = [fp, C.c_int]
        lib.orc_kdtree_destroy.argtypes = [vp]
        lib.orc_kdtree_knn_batch.argtypes = [vp, fp, C.c_int, C.c_int, C.POINTER(C.c_int64), dp, C.c_double, C.c_int]
        lib.orc_estimate_covariances.argtypes = [fp, C.c_int, C.c_int, C.c_int, dp]
        lib.orc_estimate_covariances.restype = C.c_int
        lib.orc_eig3_direct.argtypes = [dp, dp, dp]
        lib.orc_gicp_create.restype = vp
        lib.orc_gicp_create.argtypes = [fp, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_double]
        lib.orc_gicp_destroy.argtypes = [vp]
        lib.orc_gicp_linearize.argtypes = [vp, dp, C.POINTER(_Lin6)]
        lib.orc_gicp_evaluate.argtypes = [vp, dp, C.POINTER(_Lin6)]
        lib.orc_gicp_evaluate.restype = C.c_double
        lib.orc_gicp_correspondences.argtypes = [vp]
        lib.orc_gicp_correspondences.restype = C.POINTER(C.c_int64)
        lib.orc_max_threads.restype = C.c_int
        _LIB = lib
    return _LIB


def _f32(a, shape_last):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == shape_last, a.shape
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pose(T):
    """4x4 numpy (row, col) -> column-major double[16]."""
    T = np.asarray(T, dtype=np.float64)
    assert T.shape == (4, 4)
    return np.ascontiguousarray(T.T).reshape(16)


def max_threads():
    return int(_lib().orc_max_threads())


def calc_delta(T_target, T_source):
    out = np.zeros(16)
    a, b = _pose(T_target), _pose(T_source)
    _lib().orc_calc_delta(_dp(a), _dp(b), _dp(out))
    return out.reshape(4, 4).T.copy()


def expmap(xi):
    xi = np.ascontiguousarray(xi, dtype=np.float64)
    out = np.zeros(16)
    _lib().orc_pose_expmap(_dp(xi), _dp(out))
    return out.reshape(4, 4).T.copy()


def covs_as_f9(covs):
    """(N,3,3) or (N,9) -> contiguous float32 (N,9) column-major (symmetric => order-free)."""
    covs = np.asarray(covs)
    if covs.ndim == 3:
        covs = covs.transpose(0, 2, 1).reshape(len(covs), 9)
    return np.ascontiguousarray(covs, dtype=np.float32)


class OracleVoxelMap:
    """GaussianVoxelMapCPU restatement."""

    def __init__(self, resolution):
        self._h = _lib().orc_voxelmap_create(float(resolution))
        self.resolution = float(resolution)
        self._keep = []

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().orc_voxelmap_destroy(self._h)
            self._h = None

    def insert(self, points, covs, intensities=None):
        p = _f32(points, 3)
        c = covs_as_f9(covs)
        assert len(p) == len(c)
        it = np.ascontiguousarray(intensities, dtype=np.float32) if intensities is not None else None
        _lib().orc_voxelmap_insert(self._h, _fp(p), _fp(c), _fp(it), len(p))

    @property
    def num_voxels(self):
        return int(_lib().orc_voxelmap_num_voxels(self._h))

    def export(self):
        V = self.num_voxels
        coords = np.zeros((V, 3), dtype=np.int32)
        num_points = np.zeros(V, dtype=np.int32)
        means = np.zeros((V, 3))
        covs = np.zeros((V, 9))
        intens = np.zeros(V)
        _lib().orc_voxelmap_export(
            self._h,
            coords.ctypes.data_as(C.POINTER(C.c_int)),
            num_points.ctypes.data_as(C.POINTER(C.c_int)),
            _dp(means),
            _dp(covs),
            _dp(intens),
        )
        return coords, num_points, means, covs.reshape(V, 3, 3).transpose(0, 2, 1).copy(), intens

    def lookup_coord(self, c):
        return int(_lib().orc_voxelmap_lookup_coord(self._h, int(c[0]), int(c[1]), int(c[2])))

    def overlap(self, points, delta=np.eye(4)):
        p = _f32(points, 3)
        d = _pose(delta)
        return float(_lib().orc_voxelmap_overlap(self._h, _fp(p), len(p), _dp(d)))


def merge_frames(poses, frames, downsample_resolution):
    """CPU statement of merge_frames_gpu (types/gaussian_voxelmap_gpu_funcs.cu:65-152): every frame (points (N,3) f32,
    covs (N,3,3) f32, optional intensities) is transformed by its pose -- p' = R p + t (:42-50), C' = R C R^T (:52-62),
    results stored as f32 like the reference's device arrays (f64 arithmetic here, so that the only rounding is the
    final store) -- then all points go through a Gaussian voxel map at the down-sampling resolution (:122-123) whose voxel
    means / mean covariances / max intensities are the merged cloud (:125-149).
    Returns (coords (V,3) int, means (V,3) f64, covs (V,3,3) f64, intensities (V,))."""
    pts, covs, ints = [], [], []
    for T, (p, c, it) in zip(poses, frames):
        T = np.asarray(T, dtype=np.float64)
        R, t = T[:3, :3], T[:3, 3]
        pts.append((np.asarray(p, dtype=np.float64) @ R.T + t).astype(np.float32))
        covs.append(np.einsum("ij,njk,lk->nil", R, np.asarray(c, dtype=np.float64), R).astype(np.float32))
        ints.append(np.zeros(len(p), dtype=np.float32) if it is None else np.asarray(it, dtype=np.float32))  # :103-107
    m = OracleVoxelMap(downsample_resolution)
    m.insert(np.concatenate(pts), np.concatenate(covs), np.concatenate(ints))
    coords, _, means, mcovs, intens = m.export()
    return coords, means, mcovs, intens


class OracleVGICPFactor:
    """IntegratedVGICPFactor (CPU) restatement; delta = T_target^-1 T_source is passed directly."""

    def __init__(self, target: OracleVoxelMap, points, covs, num_threads=1):
        self.target = target
        self.points = _f32(points, 3)
        self.covs = covs_as_f9(covs)
        assert len(self.points) == len(self.covs)
        self._h = _lib().orc_vgicp_create(target._h, _fp(self.points), _fp(self.covs), len(self.points), int(num_threads))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().orc_vgicp_destroy(self._h)
            self._h = None

    def linearize(self, delta):
        out = _Lin6()
        d = _pose(delta)
        _lib().orc_vgicp_linearize(self._h, _dp(d), C.byref(out))
        return Linearized6.from_struct(out)

    def update_correspondences(self, delta):
        d = _pose(delta)
        _lib().orc_vgicp_update_correspondences(self._h, _dp(d))

    def evaluate(self, delta):
        out = _Lin6()
        d = _pose(delta)
        _lib().orc_vgicp_evaluate(self._h, _dp(d), C.byref(out))
        return Linearized6.from_struct(out)

    def error(self, delta):
        d = _pose(delta)
        return float(_lib().orc_vgicp_error(self._h, _dp(d)))

    @property
    def num_inliers(self):
        return int(_lib().orc_vgicp_num_inliers(self._h))

    def correspondences(self):
        p = _lib().orc_vgicp_correspondences(self._h)
        return np.ctypeslib.as_array(p, shape=(len(self.points),)).copy()


class OracleKdTree:
    def __init__(self, points):
        self.points = _f32(points, 3)
        self._h = _lib().orc_kdtree_create(_fp(self.points), len(self.points))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().orc_kdtree_destroy(self._h)
            self._h = None

    def knn(self, queries, k, max_sq_dist=np.finfo(np.float64).max, num_threads=1):
        q = _f32(queries, 3)
        idx = np.zeros((len(q), k), dtype=np.int64)
        d = np.zeros((len(q), k))
        _lib().orc_kdtree_knn_batch(self._h, _fp(q), len(q), int(k), idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(d), float(max_sq_dist), int(num_threads))
        return idx, d


def estimate_covariances(points, k=10, num_threads=1):
    """estimate_covariances(points, n, k) -> (N,3,3) double; also returns #points with < k neighbours."""
    p = _f32(points, 3)
    out = np.zeros((len(p), 9))
    short = _lib().orc_estimate_covariances(_fp(p), len(p), int(k), int(num_threads), _dp(out))
    return out.reshape(len(p), 3, 3).transpose(0, 2, 1).copy(), int(short)


def eig3_direct(mat):
    m = np.ascontiguousarray(np.asarray(mat, dtype=np.float64).T).reshape(9)
    ev = np.zeros(3)
    V = np.zeros(9)
    _lib().orc_eig3_direct(_dp(m), _dp(ev), _dp(V))
    return ev, V.reshape(3, 3).T.copy()


class OracleGICPFactor:
    def __init__(self, target_points, target_covs, points, covs, num_threads=1, max_corr_dist_sq=1.0):
        self.tp = _f32(target_points, 3)
        self.tc = covs_as_f9(target_covs)
        self.points = _f32(points, 3)
        self.covs = covs_as_f9(covs)
        self._h = _lib().orc_gicp_create(
            _fp(self.tp), _fp(self.tc), len(self.tp), _fp(self.points), _fp(self.covs), len(self.points), int(num_threads), float(max_corr_dist_sq)
        )

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().orc_gicp_destroy(self._h)
            self._h = None

    def linearize(self, delta):
        out = _Lin6()
        d = _pose(delta)
        _lib().orc_gicp_linearize(self._h, _dp(d), C.byref(out))
        return Linearized6.from_struct(out)

    def evaluate(self, delta):
        out = _Lin6()
        d = _pose(delta)
        _lib().orc_gicp_evaluate(self._h, _dp(d), C.byref(out))
        return Linearized6.from_struct(out)

    def correspondences(self):
        p = _lib().orc_gicp_correspondences(self._h)
        return np.ctypeslib.as_array(p, shape=(len(self.points),)).copy()
