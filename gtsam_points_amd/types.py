"""Host-side mirrors of the reference's GPU types, written against the C-ABI.

  PointCloudGPU        <- types/point_cloud_gpu.hpp  (device attribute arrays, float3 / 3x3 float layout)
  GaussianVoxelMapGPU  <- types/gaussian_voxelmap_gpu.hpp:38-114
  overlap_gpu          <- types/gaussian_voxelmap_gpu_funcs.cu:192-406 (single, multi-target union, pairwise batch)
  merge_frames_gpu     <- types/gaussian_voxelmap_gpu_funcs.cu:65-152

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


class OffloadableGPU:
    """types/offloadable.hpp:17-63, offloadable.cpp: a global access counter, `touch()` = remember the access and make sure
    the data is on the GPU.  Applications sort their frames by last_accessed_time() to decide what to offload."""

    _access_counter = 0

    def __init__(self):
        self._last_access = OffloadableGPU._access_counter

    @staticmethod
    def current_access_time():
        return OffloadableGPU._access_counter

    def last_accessed_time(self):
        return self._last_access

    def touch(self, stream=None):
        self._last_access = OffloadableGPU._access_counter
        OffloadableGPU._access_counter += 1
        return self.reload_gpu(stream)


class PointCloudGPU(OffloadableGPU):
    """Device attribute arrays in the reference layout (types/point_cloud.hpp:114-118):
    points_gpu float[N][3], covs_gpu float[N][9] (3x3 column-major, symmetric), normals_gpu float[N][3],
    intensities_gpu float[N].  Like the reference class (a PointCloudCPU with device mirrors) it keeps the host arrays it was
    given, so that offload_gpu() / reload_gpu() (types/point_cloud_gpu.cu:304-370) can drop and restore the device side."""

    _ATTRS = ("points", "covs", "normals", "intensities", "times")

    def __init__(self, points=None, covs=None, normals=None, intensities=None, device="cuda:0"):
        import torch

        OffloadableGPU.__init__(self)
        self.device = torch.device(device)
        self.points_gpu = None
        self.covs_gpu = None
        self.normals_gpu = None
        self.intensities_gpu = None
        self.times_gpu = None
        self.num_points = 0
        self._host = {}       # attribute -> the host array as given (None for device-only attributes)
        self.generation = 0   # bumped whenever the device arrays are re-allocated (factors re-read the pointers)
        if points is not None:
            self.add_points(points)
        if covs is not None:
            self.add_covs(covs)
        if normals is not None:
            self.add_normals(normals)
        if intensities is not None:
            self.add_intensities(intensities)

    def _forget_mirrors(self, attr):
        """The device tensor of `attr` is about to be replaced or dropped and torch may hand its address out again: the library must not let a later
        factor join a packed source mirror built from the old contents (gp_source_mirror_invalidate, include/gtsam_points_hip.h)."""
        t = getattr(self, attr + "_gpu", None)
        if t is not None and attr in ("points", "covs"):
            try:
                self._lib().gp_source_mirror_invalidate(C.c_void_p(t.data_ptr()))
            except Exception:  # interpreter shutdown
                pass

    def __del__(self):
        for a in ("points", "covs"):
            self._forget_mirrors(a)

    def _upload(self, a, width):
        import torch

        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.float32).reshape(-1, width).contiguous()
        else:
            a = np.ascontiguousarray(np.asarray(a).reshape(-1, width), dtype=np.float32)
            t = torch.from_numpy(a).to(self.device)
        return t

    def _pack_upload(self, a, matrix):
        """Raw host array of D-vectors / DxD matrices (D in {3,4}, float32 or float64, in the memory layout the
        reference's Eigen arrays have) -> float[N][3] / float[N][9] on the device; the conversion runs on the GPU
        (gp_cloud_upload_vec3 / _mat3) instead of element-wise on the host (types/point_cloud_gpu.cu:26-62,110-201)."""
        import torch

        a = np.asarray(a)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        if matrix:
            if a.ndim != 3 or a.shape[1] != a.shape[2] or a.shape[1] not in (3, 4):
                raise ValueError("covariances must be (N,3,3) or (N,4,4)")
            a = np.ascontiguousarray(a.transpose(0, 2, 1))  # Eigen is column-major
        else:
            if a.ndim != 2 or a.shape[1] not in (3, 4):
                raise ValueError("expected (N,3) or (N,4)")
            a = np.ascontiguousarray(a)
        n, dim = a.shape[0], a.shape[1]
        out = torch.empty((n, 9 if matrix else 3), dtype=torch.float32, device=self.device)
        fn = self._lib().gp_cloud_upload_mat3 if matrix else self._lib().gp_cloud_upload_vec3
        torch.cuda.current_stream(self.device).synchronize()
        _capi.check(fn(a.ctypes.data, int(a.dtype == np.float64), dim, n, C.c_void_p(out.data_ptr()), None), "gp_cloud_upload")
        return out

    @staticmethod
    def _lib():
        return _capi.load()

    @staticmethod
    def _is_tensor(a):
        import torch

        return isinstance(a, torch.Tensor)  # (numpy >= 2 arrays also carry a .device attribute)

    def add_points(self, points):  # add_points_gpu, types/point_cloud_gpu.cu:110-140 (D in {3,4})
        self._forget_mirrors("points")
        if self._is_tensor(points):
            self.points_gpu = self._upload(points[:, :3], 3)
        else:
            self.points_gpu = self._pack_upload(points, matrix=False)
            self._host["points"] = np.asarray(points)
        self.num_points = int(self.points_gpu.shape[0])
        self.generation += 1

    def add_covs(self, covs):  # add_covs_gpu: (N,3,3) or (N,4,4); (N,9) = already column-major rows
        self._forget_mirrors("covs")
        if self._is_tensor(covs):
            self.covs_gpu = self._upload(covs, 9)
            self.generation += 1
            return
        c = np.asarray(covs)
        self.covs_gpu = self._upload(c, 9) if c.ndim == 2 else self._pack_upload(c, matrix=True)
        self._host["covs"] = c
        self.generation += 1

    def add_normals(self, normals):
        if self._is_tensor(normals):
            self.normals_gpu = self._upload(normals[:, :3], 3)
        else:
            self.normals_gpu = self._pack_upload(normals, matrix=False)
            self._host["normals"] = np.asarray(normals)
        self.generation += 1

    @staticmethod
    def from_device(points_gpu, covs_gpu=None, intensities_gpu=None):
        """Adopt float32 device tensors that are already in the reference layout.
        Contract: the library treats adopted arrays as IMMUTABLE while factors use them (it may keep a packed private copy of (points, covs), keyed on their
        addresses: DESIGN 3). A caller that rewrites an adopted tensor in place calls `contents_changed()` afterwards."""
        pc = PointCloudGPU(device=str(points_gpu.device))
        pc.points_gpu = points_gpu
        pc.covs_gpu = covs_gpu
        pc.intensities_gpu = intensities_gpu
        pc.num_points = int(points_gpu.shape[0])
        pc.generation += 1
        return pc

    def contents_changed(self, attrs=("points", "covs")):
        """The caller rewrote device arrays of this cloud in place (same addresses, new contents): drop the library's packed mirrors of them and make every factor
        re-read the cloud (generation bump) -- gp_source_mirror_invalidate, include/gtsam_points_hip.h."""
        for a in attrs:
            self._forget_mirrors(a)
        self.generation += 1

    def add_intensities(self, intensities):
        self.intensities_gpu = self._upload(intensities if self._is_tensor(intensities) else np.asarray(intensities).reshape(-1, 1), 1)
        if not self._is_tensor(intensities):
            self._host["intensities"] = np.asarray(intensities)
        self.generation += 1

    def add_times(self, times):  # add_times_gpu, types/point_cloud_gpu.cu:88-105: float timestamps on the device
        self.times_gpu = self._upload(times if self._is_tensor(times) else np.asarray(times).reshape(-1, 1), 1)
        if not self._is_tensor(times):
            self._host["times"] = np.asarray(times)
        self.generation += 1

    @staticmethod
    def clone(frame):
        """PointCloudGPU::clone (types/point_cloud_gpu.cu:26-62): a deep copy.  Host attributes are uploaded again; attributes the
        source holds on the device only are copied device to device (the reference leaves those out: its TODO at :29)."""
        out = PointCloudGPU(device=str(frame.device))
        adders = {"points": out.add_points, "covs": out.add_covs, "normals": out.add_normals, "intensities": out.add_intensities, "times": out.add_times}
        for a in PointCloudGPU._ATTRS:
            if frame._host.get(a) is not None:
                adders[a](np.array(frame._host[a], copy=True))
            elif getattr(frame, a + "_gpu") is not None:
                setattr(out, a + "_gpu", getattr(frame, a + "_gpu").clone())
                out.generation += 1
        out.num_points = frame.num_points
        return out

    # ---- OffloadableGPU (types/point_cloud_gpu.cu:281-370) ----
    def loaded_on_gpu(self):
        return any(getattr(self, a + "_gpu") is not None for a in self._ATTRS)

    def memory_usage_gpu(self):
        width = {"points": 12, "covs": 36, "normals": 12, "intensities": 4, "times": 4}
        return sum(width[a] * self.num_points for a in self._ATTRS if getattr(self, a + "_gpu") is not None)

    def download(self, attr):
        """download_points_gpu / _covs_gpu / _normals_gpu / _intensities_gpu (:221-279): the float device array, on the host"""
        t = getattr(self, attr + "_gpu")
        if t is None:
            raise _capi.GPError(f"error: frame does not have {attr} on GPU!!")
        a = t.cpu().numpy()
        return a.reshape(-1, 3, 3).transpose(0, 2, 1).copy() if attr == "covs" else (a.reshape(-1) if attr in ("intensities", "times") else a)

    def offload_gpu(self, stream=None):
        """frees the device arrays (:304-336); returns False when there was nothing to offload.  An attribute that only ever
        existed on the device (from_device) is downloaded first so that reload_gpu() can restore it."""
        if not self.loaded_on_gpu():
            return False
        for a in self._ATTRS:
            if getattr(self, a + "_gpu") is not None:
                if self._host.get(a) is None:
                    self._host[a] = self.download(a)
                self._forget_mirrors(a)
                setattr(self, a + "_gpu", None)
        self.generation += 1
        return True

    def reload_gpu(self, stream=None):
        """re-uploads every attribute that has a host copy (:338-370); returns False when the cloud is already on the GPU"""
        if self.loaded_on_gpu():
            return False
        host, self._host = self._host, {}
        adders = {"points": self.add_points, "covs": self.add_covs, "normals": self.add_normals, "intensities": self.add_intensities, "times": self.add_times}
        reloaded = False
        for a in self._ATTRS:
            if host.get(a) is not None:
                adders[a](host[a])
                reloaded = True
        return reloaded

    def size(self):
        return self.num_points

    @staticmethod
    def ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None


class GaussianVoxelMapGPU(OffloadableGPU):
    """GaussianVoxelMapGPU(resolution, init_num_buckets=8192*2, max_bucket_scan_count=10,
    target_points_drop_rate=1e-3, stream=0), types/gaussian_voxelmap_gpu.hpp:51-56."""

    def __init__(self, resolution, init_num_buckets=8192 * 2, max_bucket_scan_count=10, target_points_drop_rate=1e-3, stream=None, _handle=None):
        OffloadableGPU.__init__(self)
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


def _poses_flat(poses):
    return np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).T.reshape(16) for T in poses]))


def overlap_gpu(target, source, delta=None):
    """The reference's overlap_gpu overload set (types/gaussian_voxelmap.hpp:72-140):
      overlap_gpu(target, source, T_target_source)                      -> fraction of source points in a target voxel
      overlap_gpu([targets], source, [Ts_target_source])                -> fraction in a voxel of ANY target
      overlap_gpu([targets], [sources], [Ts_target_source])             -> list of per-pair fractions (one launch)
    """
    if isinstance(target, (list, tuple)):
        targets = list(target)
        lib = _capi.load()
        handles = (C.c_void_p * len(targets))(*[t._h.value for t in targets])
        if isinstance(source, (list, tuple)):
            sources = list(source)
            if len(sources) != len(targets):
                raise _capi.GPError("error: The number of target voxelmaps and source point clouds must be the same!!")  # :342-345
            if not sources:
                return []
            for src in sources:
                if src.points_gpu is None:
                    raise _capi.GPError("error: GPU source points have not been allocated!!")
                GaussianVoxelMapGPU._sync_torch(src)
            deltas = _poses_flat(delta)
            pts = (C.c_void_p * len(sources))(*[src.points_gpu.data_ptr() for src in sources])
            ns = (C.c_int * len(sources))(*[src.size() for src in sources])
            hits = (C.c_int * len(sources))()
            _capi.check(lib.gp_voxelmap_overlap_batch(handles, pts, ns, deltas.ctypes.data, len(sources), hits, targets[0].stream), "gp_voxelmap_overlap_batch")
            return [hits[i] / float(sources[i].size()) if sources[i].size() else 0.0 for i in range(len(sources))]
        if source.points_gpu is None:
            raise _capi.GPError("error: GPU source points have not been allocated!!")
        GaussianVoxelMapGPU._sync_torch(source)
        deltas = _poses_flat(delta) if targets else np.zeros((0, 16))
        hits = C.c_int(0)
        _capi.check(
            lib.gp_voxelmap_overlap_multi(handles, deltas.ctypes.data, len(targets), source.ptr(source.points_gpu), source.size(), C.byref(hits),
                                          targets[0].stream if targets else None),
            "gp_voxelmap_overlap_multi",
        )
        return hits.value / float(source.size()) if source.size() else 0.0
    return _overlap_single(target, source, np.eye(4) if delta is None else delta)


def merge_frames_gpu(poses, frames, downsample_resolution, stream=None, target_points_drop_rate=1e-3):
    """merge_frames_gpu(poses, frames, downsample_resolution) (types/gaussian_voxelmap_gpu_funcs.cu:65-152): every frame is
    transformed by its pose, all points go through a Gaussian voxel map at the down-sampling resolution, and the voxel
    means / mean covariances / max intensities become the merged PointCloudGPU (device to device)."""
    import torch

    frames = list(frames)
    if len(frames) != len(poses) or not frames:
        raise ValueError("merge_frames_gpu: one pose per frame, at least one frame")
    for f in frames:
        if f.points_gpu is None or f.covs_gpu is None:
            raise _capi.GPError("error: GPU points/covs not allocated!!")
        GaussianVoxelMapGPU._sync_torch(f)
    lib = _capi.load()
    F = len(frames)
    pts = (C.c_void_p * F)(*[f.points_gpu.data_ptr() for f in frames])
    covs = (C.c_void_p * F)(*[f.covs_gpu.data_ptr() for f in frames])
    ints = (C.c_void_p * F)(*[(f.intensities_gpu.data_ptr() if f.intensities_gpu is not None else None) for f in frames])
    ns = (C.c_int * F)(*[f.size() for f in frames])
    flat = _poses_flat(poses)
    h = C.c_void_p()
    _capi.check(lib.gp_merge_frames(flat.ctypes.data, pts, covs, ints, ns, F, float(downsample_resolution), float(target_points_drop_rate), stream, C.byref(h)),
                "gp_merge_frames")
    vm = GaussianVoxelMapGPU(downsample_resolution, _handle=h, stream=stream)
    V = vm.voxelmap_info.num_voxels
    views = vm.views()
    dev = frames[0].device
    out_p = torch.empty((V, 3), dtype=torch.float32, device=dev)
    out_c = torch.empty((V, 9), dtype=torch.float32, device=dev)
    out_i = torch.empty((V, 1), dtype=torch.float32, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    _capi.check(lib.gp_memcpy_d2d(C.c_void_p(out_p.data_ptr()), views.voxel_means, 12 * V, stream), "gp_memcpy_d2d")
    _capi.check(lib.gp_memcpy_d2d(C.c_void_p(out_c.data_ptr()), views.voxel_covs, 36 * V, stream), "gp_memcpy_d2d")
    _capi.check(lib.gp_memcpy_d2d(C.c_void_p(out_i.data_ptr()), views.voxel_intensities, 4 * V, stream), "gp_memcpy_d2d")
    _capi.check(lib.gp_stream_synchronize(stream), "gp_stream_synchronize")
    return PointCloudGPU.from_device(out_p, out_c, out_i)


def _overlap_single(target: GaussianVoxelMapGPU, source: PointCloudGPU, delta=np.eye(4)):

    """overlap_gpu(target, source, delta): fraction of source points that fall in a target voxel
    (types/gaussian_voxelmap_gpu_funcs.cu:192-236)."""
    if source.points_gpu is None:
        raise _capi.GPError("error: GPU source points have not been allocated!!")
    GaussianVoxelMapGPU._sync_torch(source)
    hits = C.c_int(0)
    _capi.check(target._lib.gp_voxelmap_overlap(target._h, source.ptr(source.points_gpu), source.size(), _pose16(delta), C.byref(hits), target.stream), "gp_voxelmap_overlap")
    return hits.value / float(source.size()) if source.size() else 0.0
