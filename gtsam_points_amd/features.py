"""GPU counterparts of the CPU-only pieces BASELINE.json configs[4] names, over the C-ABI:

  KdTreeGPU                 <- ann/kdtree.hpp / small_kdtree.hpp (exact k-NN; here a cell-sorted point grid)
  estimate_covariances_gpu  <- features/covariance_estimation.hpp:estimate_covariances(points, k=10)
  IntegratedGICPFactorGPU   <- factors/integrated_gicp_factor.hpp (CPU-only upstream), same calc_delta / HessianFactor protocol
"""
import ctypes as C

import numpy as np

from . import _capi
from .factors import HessianFactor, LinearizedSystem6, pose_inverse
from .types import GaussianVoxelMapGPU, PointCloudGPU, _pose16


class KdTreeGPU:
    """Exact nearest-neighbour search structure over a PointCloudGPU (KdTree::knn_search semantics)."""

    def __init__(self, frame: PointCloudGPU, cell_size=0.25, stream=None, structure=0, counters=None):
        """structure: GP_TUNE_KNN_STRUCTURE of this search structure (0 default); counters: torch uint64[8] device tensor of work counters (measurement)"""
        self._lib = _capi.load()
        self.frame = frame
        GaussianVoxelMapGPU._sync_torch(frame)
        h = C.c_void_p()
        _capi.check(self._lib.gp_point_grid_create_ex(frame.ptr(frame.points_gpu), frame.size(), float(cell_size), int(structure),
                                                      C.c_void_p(counters.data_ptr()) if counters is not None else None, stream, C.byref(h)), "gp_point_grid_create_ex")
        self._h = h
        self.stream = stream

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.gp_point_grid_destroy(self._h)
            self._h = None

    def knn_search(self, queries, k, max_sq_dist=np.finfo(np.float64).max):
        """queries: (Q,3) numpy/torch -> (indices (Q,k) int32 with -1 padding, sq_dists (Q,k) float64, num_found (Q,))"""
        import torch

        q = PointCloudGPU(queries, device=self.frame.device)
        idx = torch.empty((q.size(), k), dtype=torch.int32, device=self.frame.device)
        d = torch.empty((q.size(), k), dtype=torch.float64, device=self.frame.device)
        nf = torch.empty(q.size(), dtype=torch.int32, device=self.frame.device)
        torch.cuda.current_stream(self.frame.device).synchronize()
        _capi.check(
            self._lib.gp_knn_search(self._h, q.ptr(q.points_gpu), q.size(), int(k), float(max_sq_dist), C.c_void_p(idx.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(nf.data_ptr()), self.stream),
            "gp_knn_search",
        )
        _capi.check(self._lib.gp_stream_synchronize(self.stream), "sync")
        return idx.cpu().numpy(), d.cpu().numpy(), nf.cpu().numpy()


def estimate_covariances_gpu(frame: PointCloudGPU, k_neighbors=10, cell_size=0.0, stream=None, structure=0, counters=None):
    """estimate_covariances(points, n, k): fills frame.covs_gpu (float [N][9]); returns the number of points with < k neighbours.
    structure / counters: per call, see KdTreeGPU (not part of the reference API)."""
    import torch

    lib = _capi.load()
    covs = torch.empty((frame.size(), 9), dtype=torch.float32, device=frame.device)
    GaussianVoxelMapGPU._sync_torch(frame)
    short = C.c_int(0)
    _capi.check(lib.gp_estimate_covariances_ex(frame.ptr(frame.points_gpu), frame.size(), int(k_neighbors), float(cell_size), C.c_void_p(covs.data_ptr()), C.byref(short),
                                               int(structure), C.c_void_p(counters.data_ptr()) if counters is not None else None, stream), "gp_estimate_covariances_ex")
    # the old covariance tensor goes back to torch's caching allocator, which hands same-size blocks out again: a packed source mirror keyed on its address must
    # not survive it, and factors that cached the pointer must re-read it
    frame._forget_mirrors("covs")
    frame.covs_gpu = covs
    frame._host.pop("covs", None)  # the host copy (if any) described the previous covariances
    frame.generation += 1
    return short.value


class IntegratedGICPFactorGPU:
    """GICP matching-cost factor on the GPU: 1-NN correspondences within max_correspondence_distance (default 1 m,
    integrated_gicp_factor_impl.hpp:30), then the same residual / Jacobian algebra as VGICP."""

    def __init__(self, target_key, source_key, target: PointCloudGPU, source: PointCloudGPU, max_correspondence_distance=1.0, stream=None, _fixed_target_pose=None,
                 structure=0, counters=None):
        self._lib = _capi.load()
        self.is_binary = _fixed_target_pose is None
        self._keys = [target_key, source_key] if self.is_binary else [source_key]
        self.fixed_target_pose = np.eye(4) if self.is_binary else np.asarray(_fixed_target_pose, dtype=np.float64)
        for fr, what in [(target, "target"), (source, "source")]:
            if fr.points_gpu is None or fr.covs_gpu is None:
                raise _capi.GPError(f"error: {what} frame doesn't have required attributes for gicp")
        self.target, self.source = target, source
        GaussianVoxelMapGPU._sync_torch(source)
        h = C.c_void_p()
        _capi.check(
            self._lib.gp_gicp_factor_create_ex(
                target.ptr(target.points_gpu), target.ptr(target.covs_gpu), target.size(), source.ptr(source.points_gpu), source.ptr(source.covs_gpu), source.size(),
                float(max_correspondence_distance) ** 2, int(structure), C.c_void_p(counters.data_ptr()) if counters is not None else None, stream, C.byref(h),
            ),
            "gp_gicp_factor_create_ex",
        )
        self._h = h
        self.linearization_point = np.eye(4)
        self._num_inliers = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.gp_gicp_factor_destroy(self._h)
            self._h = None

    def keys(self):
        return self._keys

    def calc_delta(self, values):
        if not self.is_binary:
            return pose_inverse(self.fixed_target_pose) @ np.asarray(values[self._keys[0]], dtype=np.float64)
        return pose_inverse(values[self._keys[0]]) @ np.asarray(values[self._keys[1]], dtype=np.float64)

    def linearize_delta(self, delta):
        rec = _capi.Linearized6()
        _capi.check(self._lib.gp_gicp_factor_linearize(self._h, _pose16(delta), C.byref(rec)), "gp_gicp_factor_linearize")
        l = LinearizedSystem6(rec)
        self._num_inliers = l.num_inliers
        self.linearization_point = np.asarray(delta, dtype=np.float64)
        self._linearized = True
        return l

    def linearize(self, values):
        l = self.linearize_delta(self.calc_delta(values))
        if self.is_binary:
            return HessianFactor(self._keys, {(0, 0): l.H_target, (0, 1): l.H_target_source, (1, 1): l.H_source}, [-l.b_target, -l.b_source], l.error)
        return HessianFactor(self._keys, {(0, 0): l.H_source}, [-l.b_source], l.error)

    def error(self, values):
        """evaluate(delta) on the correspondences of the last linearise; without one they are computed at `delta` itself first
        (integrated_gicp_factor_impl.hpp:226-228: update_correspondences(delta) when none are stored)"""
        delta = self.calc_delta(values)
        if not getattr(self, "_linearized", False):
            self.linearization_point = np.asarray(delta, dtype=np.float64)
            self._linearized = True
        out = C.c_double()
        _capi.check(self._lib.gp_gicp_factor_compute_error(self._h, _pose16(self.linearization_point), _pose16(delta), C.byref(out)), "gp_gicp_factor_compute_error")
        return out.value

    def num_inliers(self):
        return self._num_inliers
