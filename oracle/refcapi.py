"""ctypes bindings of oracle/_ref/libref.so: the REFERENCE's own CPU VGICP sources (compiled from /root/reference by
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
        lib.ref_vgicp_create.restype = vp
        lib.ref_vgicp_create.argtypes = [vp, fp, fp, C.c_int, C.c_int]
        lib.ref_vgicp_destroy.argtypes = [vp]
        lib.ref_vgicp_linearize.argtypes = [vp, dp, C.POINTER(_Lin6)]
        lib.ref_vgicp_error.argtypes = [vp, dp]
        lib.ref_vgicp_error.restype = C.c_double
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
