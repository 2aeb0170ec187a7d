"""gtsam_points_amd -- MI355X (gfx950) native VGICP path behind koide3/gtsam_points' GPU API.

The compute lives in libgtsam_points_hip.so (hand-written HIP, C-ABI in include/gtsam_points_hip.h).
This package is the thin host-side mirror of the reference's classes used by the tests and bench;
gtsam_points_amd/host/ holds the C++ mirror a GTSAM application links.  There is NO CPU fallback:
every class raises if the HIP library cannot be loaded.
"""
from ._capi import GPError, LIB_PATH, load  # noqa: F401
from .factors import (  # noqa: F401
    HessianFactor,
    IntegratedVGICPFactorGPU,
    LinearizationHook,
    LinearizedSystem6,
    NonlinearFactorGPU,
    NonlinearFactorSetGPU,
    StreamTempBufferRoundRobin,
    TempBufferManager,
    create_nonlinear_factor_set_gpu,
    pose_inverse,
)
from .features import IntegratedGICPFactorGPU, KdTreeGPU, estimate_covariances_gpu  # noqa: F401
from .solver import DenseLinearSystemGPU, LevenbergMarquardtGraphGPU, SparseLinearSystemGPU, linearize_on_device, sparse_symbolic  # noqa: F401
from .types import GaussianVoxelMapGPU, PointCloudGPU, merge_frames_gpu, overlap_gpu  # noqa: F401

__all__ = [
    "GPError",
    "GaussianVoxelMapGPU",
    "HessianFactor",
    "IntegratedGICPFactorGPU",
    "IntegratedVGICPFactorGPU",
    "KdTreeGPU",
    "estimate_covariances_gpu",
    "LinearizationHook",
    "LinearizedSystem6",
    "NonlinearFactorGPU",
    "NonlinearFactorSetGPU",
    "PointCloudGPU",
    "StreamTempBufferRoundRobin",
    "TempBufferManager",
    "create_nonlinear_factor_set_gpu",
    "overlap_gpu",
    "merge_frames_gpu",
    "DenseLinearSystemGPU",
    "LevenbergMarquardtGraphGPU",
    "SparseLinearSystemGPU",
    "sparse_symbolic",
    "linearize_on_device",
    "load",
]
