"""ctypes bindings of oracle/_ref/libref.so: the REFERENCE's own CPU VGICP / kd-tree / covariance / GICP sources (compiled from /root/reference by
oracle/ref_shim/Makefile against stand-in Eigen/GTSAM headers).  TEST INFRASTRUCTURE ONLY.  Used to pin vgicp_oracle.c."""
import ctypes as C
import os
import subprocess

import numpy as np

from .capi import Linearized6, _Lin6, _f32, _fp, _dp, _pose, covs_as_f9

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBREF = os.path.join(_HERE, "_ref", "libref.so")
_LIB = None


def available():
    return os.path.exists(LIBREF)


def build():
    """Build oracle/_ref/libref.so when the reference tree is mounted (no-op otherwise)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref_shim"), "-s"])
    return available()


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(LIBREF)
        vp, dp, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
        lib.ref_voxelmap_create.restype = vp
        lib.ref_voxelmap_create.argtypes = [C.c_double]
        lib.ref_voxelmap_destroy.argtypes = [vp]
        lib.ref_voxelmap_insert.argtypes = [vp, fp, fp, C.c_int]
        lib.ref_voxelmap_num_voxels.argtypes = [vp]
        lib.ref_voxelmap_num_voxels.restype = C.c_int
        lib.ref_voxelmap_save_compact.argtypes = [vp, C.c_char_p]
        lib.ref_voxelmap_load.restype = vp
        lib.ref_voxelmap_load.argtypes = [C.c_char_p]
        lib.ref_voxelmap_resolution.restype = C.c_double
        lib.ref_voxelmap_resolution.argtypes = [vp]
        lib.ref_voxelmap_export.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_voxelmap_overlap.restype = C.c_double
        lib.ref_voxelmap_overlap.argtypes = [vp, fp, C.c_int, dp]
        lib.ref_vgicp_create.restype = vp
        lib.ref_vgicp_create.argtypes = [vp, fp, fp, C.c_int, C.c_int]
        lib.ref_vgicp_destroy.argtypes = [vp]
        lib.ref_vgicp_linearize.argtypes = [vp, dp, C.POINTER(_Lin6)]
        lib.ref_vgicp_error.argtypes = [vp, dp]
        lib.ref_vgicp_error.restype = C.c_double
        lib.ref_kdtree_create.restype = vp
        lib.ref_kdtree_create.argtypes = [fp, C.c_int]
        lib.ref_kdtree_destroy.argtypes = [vp]
        lib.ref_kdtree_knn.argtypes = [vp, dp, C.c_int, C.POINTER(C.c_longlong), dp, C.c_double]
        lib.ref_kdtree_knn.restype = C.c_int
        lib.ref_estimate_covariances.argtypes = [fp, C.c_int, C.c_int, C.c_int, dp]
        lib.ref_gicp_create.restype = vp
        lib.ref_gicp_create.argtypes = [fp, fp, C.c_int, fp, fp, C.c_int, C.c_int, C.c_double]
        lib.ref_gicp_destroy.argtypes = [vp]
        lib.ref_gicp_linearize.argtypes = [vp, dp, C.POINTER(_Lin6)]
        _LIB = lib
    return _LIB


class RefVoxelMap:
    """gtsam_points::GaussianVoxelMapCPU (the reference's own class)"""

    def __init__(self, resolution):
        self._h = _lib().ref_voxelmap_create(float(resolution))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().ref_voxelmap_destroy(self._h)
            self._h = None

    def insert(self, points, covs):
        p, c = _f32(points, 3), covs_as_f9(covs)
        _lib().ref_voxelmap_insert(self._h, _fp(p), _fp(c), len(p))

    @property
    def num_voxels(self):
        return int(_lib().ref_voxelmap_num_voxels(self._h))

    # ---- the reference's own on-disk format (GaussianVoxelMapCPU::save_compact / load) and overlap() ----
    def save_compact(self, path):
        _lib().ref_voxelmap_save_compact(self._h, str(path).encode())

    @staticmethod
    def load(path):
        h = _lib().ref_voxelmap_load(str(path).encode())
        if not h:
            return None
        m = RefVoxelMap.__new__(RefVoxelMap)
        m._h = h
        return m

    def voxel_resolution(self):
        return float(_lib().ref_voxelmap_resolution(self._h))

    def export(self):
        V = self.num_voxels
        coords, npts = np.zeros((V, 3), np.int32), np.zeros(V, np.int32)
        means, covs, ints = np.zeros((V, 3)), np.zeros((V, 9)), np.zeros(V)
        _lib().ref_voxelmap_export(self._h, coords.ctypes.data, npts.ctypes.data, means.ctypes.data, covs.ctypes.data, ints.ctypes.data)
        return coords, npts, means, covs.reshape(V, 3, 3).transpose(0, 2, 1).copy(), ints

    def overlap(self, points, delta=np.eye(4)):
        p = _f32(points, 3)
        d = np.ascontiguousarray(np.asarray(delta, dtype=np.float64).T).reshape(16).copy()
        return float(_lib().ref_voxelmap_overlap(self._h, _fp(p), len(p), d.ctypes.data_as(C.POINTER(C.c_double))))


class RefVGICPFactor:
    """gtsam_points::IntegratedVGICPFactor (the reference's own class), keys (0, 1), pose 0 = identity, pose 1 = delta"""

    def __init__(self, target: RefVoxelMap, points, covs, num_threads=1):
        self.target = target
        self.points, self.covs = _f32(points, 3), covs_as_f9(covs)
        self._h = _lib().ref_vgicp_create(target._h, _fp(self.points), _fp(self.covs), len(self.points), int(num_threads))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().ref_vgicp_destroy(self._h)
            self._h = None

    def linearize(self, delta):
        out = _Lin6()
        d = _pose(delta)
        _lib().ref_vgicp_linearize(self._h, _dp(d), C.byref(out))
        return Linearized6.from_struct(out)

    def error(self, delta):
        d = _pose(delta)
        return float(_lib().ref_vgicp_error(self._h, _dp(d)))


class RefKdTree:
    """gtsam_points::KdTree (ann/kdtree.cpp over small_kdtree.hpp / knn_result.hpp, the reference's own code)"""

    def __init__(self, points):
        self.points = _f32(points, 3)
        self._h = _lib().ref_kdtree_create(_fp(self.points), len(self.points))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().ref_kdtree_destroy(self._h)
            self._h = None

    def knn(self, queries, k, max_sq_dist=np.finfo(np.float64).max):
        q = np.ascontiguousarray(_f32(queries, 3), dtype=np.float64)
        idx = np.full((len(q), k), -1, dtype=np.int64)
        d = np.zeros((len(q), k))
        found = np.zeros(len(q), dtype=np.int32)
        for i in range(len(q)):
            found[i] = _lib().ref_kdtree_knn(self._h, _dp(q[i]), int(k), idx[i].ctypes.data_as(C.POINTER(C.c_longlong)), _dp(d[i]), float(max_sq_dist))
        return idx, d, found


def ref_estimate_covariances(points, k=10, num_threads=1):
    """gtsam_points::estimate_covariances (features/covariance_estimation.cpp, the reference's own code) -> (N,3,3) double.
    NOTE: the SelfAdjointEigenSolver underneath is the stand-in's Jacobi solver, not Eigen's closed form."""
    p = _f32(points, 3)
    out = np.zeros((len(p), 9))
    _lib().ref_estimate_covariances(_fp(p), len(p), int(k), int(num_threads), _dp(out))
    return out.reshape(len(p), 3, 3).transpose(0, 2, 1).copy()


class RefGICPFactor:
    """gtsam_points::IntegratedGICPFactor_<PointCloud, PointCloud> (the reference's own class), keys (0, 1)"""

    def __init__(self, target_points, target_covs, points, covs, num_threads=1, max_corr_dist_sq=1.0):
        self.tp, self.tc = _f32(target_points, 3), covs_as_f9(target_covs)
        self.points, self.covs = _f32(points, 3), covs_as_f9(covs)
        self._h = _lib().ref_gicp_create(_fp(self.tp), _fp(self.tc), len(self.tp), _fp(self.points), _fp(self.covs), len(self.points), int(num_threads),
                                         float(max_corr_dist_sq))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().ref_gicp_destroy(self._h)
            self._h = None

    def linearize(self, delta):
        out = _Lin6()
        d = _pose(delta)
        _lib().ref_gicp_linearize(self._h, _dp(d), C.byref(out))
        return Linearized6.from_struct(out)
