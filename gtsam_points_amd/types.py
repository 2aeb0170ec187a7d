"""Host-side mirrors of the reference's GPU types, written against the C-ABI.

  PointCloudGPU        <- types/point_cloud_gpu.hpp  (device attribute arrays, float3 / 3x3 float layout)
  GaussianVoxelMapGPU  <- types/gaussian_voxelmap_gpu.hpp:38-114
  overlap_gpu          <- types/gaussian_voxelmap_gpu_funcs.cu:192-236

torch is used only as the device allocator / stream provider (plumbing).
"""
import ctypes as C

import numpy as np

from . import _capi


def _pose16(T):
    """4x4 (row, col) numpy -> column-major double[16] ctypes array."""
    T = np.asarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("pose must be 4x4")
    flat = np.ascontiguousarray(T.T).reshape(16)
    return (C.c_double * 16)(*flat.tolist())


class PointCloudGPU:
    """Device attribute arrays in the reference layout (types/point_cloud.hpp:114-118):
    points_gpu float[N][3], covs_gpu float[N][9] (3x3 column-major, symmetric), normals_gpu float[N][3],
    intensities_gpu float[N]."""

    def __init__(self, points=None, covs=None, normals=None, intensities=None, device="cuda:0"):
        import torch

        self.device = torch.device(device)
        self.points_gpu = None
        self.covs_gpu = None
        self.normals_gpu = None
        self.intensities_gpu = None
        self.num_points = 0
        if points is not None:
            self.add_points(points)
        if covs is not None:
            self.add_covs(covs)
        if normals is not None:
            self.add_normals(normals)
        if intensities is not None:
            self.add_intensities(intensities)

    def _upload(self, a, width):
        import torch

        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.float32).reshape(-1, width).contiguous()
        else:
            a = np.ascontiguousarray(np.asarray(a).reshape(-1, width), dtype=np.float32)
            t = torch.from_numpy(a).to(self.device)
        return t

    def add_points(self, points):  # add_points_gpu, types/point_cloud_gpu.cu:110-140 (D in {3,4})
        p = np.asarray(points) if not hasattr(points, "device") else points
        if hasattr(p, "shape") and p.shape[-1] == 4:
            p = p[:, :3]
        self.points_gpu = self._upload(p, 3)
        self.num_points = int(self.points_gpu.shape[0])

    def add_covs(self, covs):  # add_covs_gpu: (N,3,3) or (N,4,4) or (N,9)
        c = covs
        if not hasattr(c, "device"):
            c = np.asarray(c)
            if c.ndim == 3 and c.shape[1] == 4:
                c = c[:, :3, :3]
            if c.ndim == 3:
                c = c.transpose(0, 2, 1)  # column-major storage
            c = c.reshape(len(c), 9)
        self.covs_gpu = self._upload(c, 9)

    def add_normals(self, normals):
        n = normals if hasattr(normals, "device") else np.asarray(normals)[:, :3]
        self.normals_gpu = self._upload(n, 3)

    def add_intensities(self, intensities):
        self.intensities_gpu = self._upload(np.asarray(intensities).reshape(-1, 1) if not hasattr(intensities, "device") else intensities, 1)

    def size(self):
        return self.num_points

    @staticmethod
    def ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None


class GaussianVoxelMapGPU:
    """GaussianVoxelMapGPU(resolution, init_num_buckets=8192*2, max_bucket_scan_count=10,
    target_points_drop_rate=1e-3, stream=0), types/gaussian_voxelmap_gpu.hpp:51-56."""

    def __init__(self, resolution, init_num_buckets=8192 * 2, max_bucket_scan_count=10, target_points_drop_rate=1e-3, stream=None, _handle=None):
        self._lib = _capi.load()
        self.stream = stream
        if _handle is not None:
            self._h = _handle
        else:
            h = C.c_void_p()
            _capi.check(
                self._lib.gp_voxelmap_create(float(resolution), int(init_num_buckets), int(max_bucket_scan_count), float(target_points_drop_rate), stream, C.byref(h)),
                "gp_voxelmap_create",
            )
            self._h = h
        self._frame = None  # keep the inserted frame alive like the reference's callers do

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.gp_voxelmap_destroy(h)
            self._h = None

    def voxel_resolution(self):
        return float(self._lib.gp_voxelmap_resolution(self._h))

    def insert(self, frame: PointCloudGPU):
        if frame.points_gpu is None or frame.covs_gpu is None:
            # the reference abort()s here (gaussian_voxelmap_gpu.cu:212-215)
            raise _capi.GPError("error: GPU points/covs not allocated!!")
        self._sync_torch(frame)
        _capi.check(
            self._lib.gp_voxelmap_insert(self._h, frame.ptr(frame.points_gpu), frame.ptr(frame.covs_gpu), frame.ptr(frame.intensities_gpu), frame.size()),
            "gp_voxelmap_insert",
        )

    @staticmethod
    def _sync_torch(frame):
        import torch

        torch.cuda.current_stream(frame.device).synchronize()

    @property
    def voxelmap_info(self):
        info = _capi.VoxelMapInfo()
        _capi.check(self._lib.gp_voxelmap_info_get(self._h, C.byref(info)), "gp_voxelmap_info_get")
        return info

    def views(self):
        v = _capi.VoxelMapViews()
        _capi.check(self._lib.gp_voxelmap_views_get(self._h, C.byref(v)), "gp_voxelmap_views_get")
        return v

    # download_buckets / download_voxel_num_points / _means / _covs / _intensities (gaussian_voxelmap_gpu.hpp:110-114)
    def download(self):
        info = self.voxelmap_info
        V, B = info.num_voxels, info.num_buckets
        buckets = np.zeros((B, 4), dtype=np.int32)
        num_points = np.zeros(V, dtype=np.int32)
        means = np.zeros((V, 3), dtype=np.float32)
        covs = np.zeros((V, 9), dtype=np.float32)
        intens = np.zeros(V, dtype=np.float32)
        _capi.check(
            self._lib.gp_voxelmap_download(self._h, buckets.ctypes.data, num_points.ctypes.data, means.ctypes.data, covs.ctypes.data, intens.ctypes.data),
            "gp_voxelmap_download",
        )
        return dict(buckets=buckets, num_points=num_points, means=means, covs=covs.reshape(V, 3, 3).transpose(0, 2, 1).copy(), intensities=intens)

    def download_f64(self):
        V = self.voxelmap_info.num_voxels
        coords = np.zeros((V, 3), dtype=np.int32)
        num_points = np.zeros(V, dtype=np.int32)
        means = np.zeros((V, 3))
        covs = np.zeros((V, 9))
        _capi.check(self._lib.gp_voxelmap_download_f64(self._h, coords.ctypes.data, num_points.ctypes.data, means.ctypes.data, covs.ctypes.data), "gp_voxelmap_download_f64")
        return coords, num_points, means, covs.reshape(V, 3, 3).transpose(0, 2, 1).copy()

    def save_compact(self, path):
        _capi.check(self._lib.gp_voxelmap_save_compact(self._h, str(path).encode()), "gp_voxelmap_save_compact")

    @staticmethod
    def load(path, stream=None):
        lib = _capi.load()
        h = C.c_void_p()
        rc = lib.gp_voxelmap_load(str(path).encode(), stream, C.byref(h))
        if rc != 0:
            return None  # the reference returns nullptr (gaussian_voxelmap_gpu.cu:374-377)
        res = lib.gp_voxelmap_resolution(h)
        return GaussianVoxelMapGPU(res, _handle=h, stream=stream)

    def memory_usage_gpu(self):
        return int(self._lib.gp_voxelmap_memory_usage_gpu(self._h))

    def loaded_on_gpu(self):
        return bool(self._lib.gp_voxelmap_loaded_on_gpu(self._h))

    def offload_gpu(self, stream=None):
        return self._lib.gp_voxelmap_offload(self._h, stream) == 0

    def reload_gpu(self, stream=None):
        return self._lib.gp_voxelmap_reload(self._h, stream) == 0

    def lookup(self, frame: PointCloudGPU, delta=np.eye(4), surface_validation=False):
        import torch

        self._sync_torch(frame)
        out = torch.empty(frame.size(), dtype=torch.int32, device=frame.device)
        normals = frame.ptr(frame.normals_gpu) if surface_validation else None
        _capi.check(
            self._lib.gp_voxelmap_lookup(self._h, frame.ptr(frame.points_gpu), normals, frame.size(), _pose16(delta), C.c_void_p(out.data_ptr()), self.stream),
            "gp_voxelmap_lookup",
        )
        return out.cpu().numpy()


def overlap_gpu(target: GaussianVoxelMapGPU, source: PointCloudGPU, delta=np.eye(4)):
    """overlap_gpu(target, source, delta): fraction of source points that fall in a target voxel
    (types/gaussian_voxelmap_gpu_funcs.cu:192-236)."""
    if source.points_gpu is None:
        raise _capi.GPError("error: GPU source points have not been allocated!!")
    GaussianVoxelMapGPU._sync_torch(source)
    hits = C.c_int(0)
    _capi.check(target._lib.gp_voxelmap_overlap(target._h, source.ptr(source.points_gpu), source.size(), _pose16(delta), C.byref(hits), target.stream), "gp_voxelmap_overlap")
    return hits.value / float(source.size()) if source.size() else 0.0
