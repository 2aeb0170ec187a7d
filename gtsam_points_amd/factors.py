"""Host-side mirrors of the reference's GPU factor API, written against the C-ABI.

  NonlinearFactorGPU         <- factors/nonlinear_factor_gpu.hpp:49-121 (the 10 virtuals)
  IntegratedVGICPFactorGPU   <- factors/integrated_vgicp_factor_gpu.{hpp,cpp}
  NonlinearFactorSetGPU      <- cuda/nonlinear_factor_set_gpu.{hpp,cpp}
  LinearizationHook          <- optimizers/linearization_hook.{hpp,cpp}
  StreamTempBufferRoundRobin <- cuda/stream_temp_buffer_roundrobin.hpp:49-65

GTSAM is not on this image, so `Values` is a dict {key: 4x4 double pose} and linearize() returns a
small HessianFactor stand-in carrying the blocks in GTSAM's HessianFactor order
(G11, G12, g1, G22, g2, f) = (H_target, H_target_source, -b_target, H_source, -b_source, error)
(integrated_vgicp_factor_gpu.cpp:199-213).
"""
import ctypes as C
import sys

import numpy as np

from . import _capi
from .types import GaussianVoxelMapGPU, PointCloudGPU, _pose16


def pose_inverse(T):
    """gtsam::Pose3::inverse(): (R^T, -R^T t)"""
    T = np.asarray(T, dtype=np.float64)
    out = np.eye(4)
    out[:3, :3] = T[:3, :3].T
    out[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return out


def col_major_6x6(a):
    return np.frombuffer(a, dtype=np.float64).reshape(6, 6).T.copy()


class LinearizedSystem6:
    """LinearizedSystem6 (cuda/kernels/linearized_system.cuh:10-71) as numpy, double."""

    def __init__(self, rec: _capi.Linearized6):
        self.num_inliers = int(round(rec.num_inliers))
        self.error = float(rec.error)
        self.H_target = col_major_6x6(rec.H_target)
        self.H_source = col_major_6x6(rec.H_source)
        self.H_target_source = col_major_6x6(rec.H_target_source)
        self.b_target = np.frombuffer(rec.b_target, dtype=np.float64).copy()
        self.b_source = np.frombuffer(rec.b_source, dtype=np.float64).copy()

    @staticmethod
    def from_doubles(d):
        rec = _capi.Linearized6.from_buffer_copy(np.ascontiguousarray(d, dtype=np.float64).tobytes())
        return LinearizedSystem6(rec)


class HessianFactor:
    """Minimal stand-in for gtsam::HessianFactor (unary or binary), blocks in GTSAM order."""

    def __init__(self, keys, G, g, f):
        self.keys = list(keys)
        self.G = G  # dict (i, j) -> 6x6 for i <= j
        self.g = g  # list of 6-vectors
        self.f = float(f)

    def information(self):
        n = len(self.keys)
        H = np.zeros((6 * n, 6 * n))
        for (i, j), blk in self.G.items():
            H[6 * i : 6 * i + 6, 6 * j : 6 * j + 6] = blk
            if i != j:
                H[6 * j : 6 * j + 6, 6 * i : 6 * i + 6] = blk.T
        return H

    def linear_term(self):
        return np.concatenate(self.g)


class NonlinearFactorGPU:
    """factors/nonlinear_factor_gpu.hpp:49-121"""

    def __init__(self, keys):
        self._keys = list(keys)

    def keys(self):
        return self._keys

    def linearization_input_size(self):
        raise NotImplementedError

    def linearization_output_size(self):
        raise NotImplementedError

    def evaluation_input_size(self):
        raise NotImplementedError

    def evaluation_output_size(self):
        raise NotImplementedError

    def set_linearization_point(self, values, lin_input_cpu):
        raise NotImplementedError

    def issue_linearize(self, lin_input_cpu, lin_input_gpu, lin_output_gpu):
        raise NotImplementedError

    def store_linearized(self, lin_output_cpu):
        raise NotImplementedError

    def set_evaluation_point(self, values, eval_input_cpu):
        raise NotImplementedError

    def issue_compute_error(self, lin_input_cpu, eval_input_cpu, lin_input_gpu, eval_input_gpu, eval_output_gpu):
        raise NotImplementedError

    def store_computed_error(self, eval_output_cpu):
        raise NotImplementedError

    def sync(self):
        raise NotImplementedError


class TempBufferManager:
    """cuda/stream_temp_buffer_roundrobin.hpp:19-44"""

    def __init__(self, init_buffer_size=0, _handle=None, _owned=True):
        self._lib = _capi.load()
        self._owned = _owned
        if _handle is not None:
            self._h = _handle
        else:
            h = C.c_void_p()
            _capi.check(self._lib.gp_temp_buffer_create(int(init_buffer_size), C.byref(h)), "gp_temp_buffer_create")
            self._h = h

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_h", None):
            self._lib.gp_temp_buffer_destroy(self._h)
            self._h = None

    def get_buffer(self, size):
        p = C.c_void_p()
        _capi.check(self._lib.gp_temp_buffer_get(self._h, int(size), C.byref(p)), "gp_temp_buffer_get")
        return p.value

    def clear(self):
        self._lib.gp_temp_buffer_clear(self._h)

    def clear_all(self):
        self._lib.gp_temp_buffer_clear_all(self._h)


class StreamTempBufferRoundRobin:
    """StreamTempBufferRoundRobin(num_streams=4, init_buffer_size=512 KiB), stream_temp_buffer_roundrobin.hpp:49-65"""

    def __init__(self, num_streams=4, init_buffer_size=512 * 1024):
        self._lib = _capi.load()
        h = C.c_void_p()
        _capi.check(self._lib.gp_stream_pool_create(int(num_streams), int(init_buffer_size), C.byref(h)), "gp_stream_pool_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.gp_stream_pool_destroy(self._h)
            self._h = None

    def get_stream_buffer(self):
        s, b = C.c_void_p(), C.c_void_p()
        _capi.check(self._lib.gp_stream_pool_get(self._h, C.byref(s), C.byref(b)), "gp_stream_pool_get")
        return s, TempBufferManager(_handle=b, _owned=False)

    def sync_all(self):
        _capi.check(self._lib.gp_stream_pool_sync_all(self._h), "gp_stream_pool_sync_all")

    def clear(self):
        self._lib.gp_stream_pool_clear(self._h)

    def clear_all(self):
        self._lib.gp_stream_pool_clear_all(self._h)


class IntegratedVGICPFactorGPU(NonlinearFactorGPU):
    """IntegratedVGICPFactorGPU (factors/integrated_vgicp_factor_gpu.hpp:43-170).

    Binary:  IntegratedVGICPFactorGPU(target_key, source_key, target, source, stream=None, temp_buffer=None)
    Unary :  IntegratedVGICPFactorGPU.unary(fixed_target_pose, source_key, target, source, ...)
    """

    def __init__(self, target_key, source_key, target, source, stream=None, temp_buffer=None, _fixed_target_pose=None):
        is_binary = _fixed_target_pose is None
        super().__init__([target_key, source_key] if is_binary else [source_key])
        self._lib = _capi.load()
        self.is_binary = is_binary
        self.fixed_target_pose = np.eye(4) if is_binary else np.asarray(_fixed_target_pose, dtype=np.float64)
        # the reference abort()s on these three (integrated_vgicp_factor_gpu.cpp:33-46)
        if source is None or source.points_gpu is None:
            raise _capi.GPError("error: GPU source points have not been allocated!!")
        if source.covs_gpu is None:
            raise _capi.GPError("error: GPU source covs have not been allocated!!")
        if not isinstance(target, GaussianVoxelMapGPU):
            raise _capi.GPError("error: GPU target voxels have not been created!!")
        self.target = target
        self.source = source
        self._temp_buffer = temp_buffer
        GaussianVoxelMapGPU._sync_torch(source)
        h = C.c_void_p()
        _capi.check(
            self._lib.gp_vgicp_factor_create(
                target._h,
                source.ptr(source.points_gpu),
                source.ptr(source.covs_gpu),
                source.ptr(source.normals_gpu),
                source.size(),
                stream,
                temp_buffer._h if temp_buffer is not None else None,
                C.byref(h),
            ),
            "gp_vgicp_factor_create",
        )
        self._h = h
        self._enable_offloading = False
        self._source_generation = source.generation
        self.linearized = False
        self.linearization_point = np.eye(4)
        self.evaluation_result = None
        self.linearization_result = None
        self._num_inliers = 0

    @staticmethod
    def unary(fixed_target_pose, source_key, target, source, stream=None, temp_buffer=None):
        return IntegratedVGICPFactorGPU(None, source_key, target, source, stream, temp_buffer, _fixed_target_pose=fixed_target_pose)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.gp_vgicp_factor_destroy(self._h)
            self._h = None

    # ---- reference API ----
    def print(self, s="", file=sys.stdout):
        if self.is_binary:
            print(f"{s}IntegratedVGICPFactorGPU({self._keys[0]}, {self._keys[1]})", file=file)
        else:
            print(f"{s}IntegratedVGICPFactorGPU(fixed, {self._keys[0]})", file=file)
        print(f"target_resolusion={self.target.voxel_resolution()}, |source|={self.source.size()}pts", file=file)

    def dim(self):
        return 6

    def memory_usage_gpu(self):
        return 128 + 4  # pose + count; no inlier index list exists in this implementation

    def set_tuning(self, key, value):
        """gp_vgicp_factor_set_tuning: a GP_TUNE_* knob of THIS factor's own batch of one (kernel family, source-stream policy, ...);
        nothing process-global.  Not part of the reference API."""
        _capi.check(self._lib.gp_vgicp_factor_set_tuning(self._h, int(key), int(value)), "gp_vgicp_factor_set_tuning")
        return self

    def set_enable_surface_validation(self, enable):
        _capi.check(self._lib.gp_vgicp_factor_set_surface_validation(self._h, int(bool(enable))), "set_enable_surface_validation")

    def set_inlier_update_thresh(self, trans, angle):
        _capi.check(self._lib.gp_vgicp_factor_set_inlier_update_thresh(self._h, float(trans), float(angle)), "set_inlier_update_thresh")

    def set_enable_offloading(self, enable):
        """integrated_vgicp_factor_gpu.cpp:102-104: when set, target and source are touch()ed (= reloaded if an application
        offloaded them) before every linearisation"""
        self._enable_offloading = bool(enable)

    def touch_points(self):
        """IntegratedVGICPDerivatives::touch_points (integrated_vgicp_derivatives.cu:63-78)"""
        if self._enable_offloading:
            self.target.touch(self.target.stream)
            self.source.touch()
        if self.source.generation != self._source_generation:
            if self.source.points_gpu is None or self.source.covs_gpu is None:
                raise _capi.GPError("error: GPU source points have not been allocated!!")
            GaussianVoxelMapGPU._sync_torch(self.source)
            _capi.check(
                self._lib.gp_vgicp_factor_set_source(self._h, self.source.ptr(self.source.points_gpu), self.source.ptr(self.source.covs_gpu),
                                                     self.source.ptr(self.source.normals_gpu)),
                "gp_vgicp_factor_set_source",
            )
            self._source_generation = self.source.generation

    def num_inliers(self):
        return self._num_inliers

    def inlier_fraction(self):
        return self._num_inliers / float(self.source.size())

    def get_target(self):
        return self.target

    def get_fixed_target_pose(self):
        return self.fixed_target_pose

    def clone(self):
        if self.is_binary:
            return IntegratedVGICPFactorGPU(self._keys[0], self._keys[1], self.target, self.source, None, None)
        return IntegratedVGICPFactorGPU.unary(self.fixed_target_pose, self._keys[0], self.target, self.source, None, None)

    def calc_delta(self, values):
        """integrated_vgicp_factor_gpu.cpp:152-164, kept in double"""
        if not self.is_binary:
            return pose_inverse(self.fixed_target_pose) @ np.asarray(values[self._keys[0]], dtype=np.float64)
        return pose_inverse(values[self._keys[0]]) @ np.asarray(values[self._keys[1]], dtype=np.float64)

    def error(self, values):
        """integrated_vgicp_factor_gpu.cpp:166-183"""
        if self.evaluation_result is not None:
            err = self.evaluation_result
            self.evaluation_result = None
            return err
        print("warning: computing error in sync mode seriously affects the processing speed!!", file=sys.stderr)
        if not self.linearized:
            self.linearize(values)
        out = C.c_double(0.0)
        _capi.check(
            self._lib.gp_vgicp_factor_compute_error(self._h, _pose16(self.linearization_point), _pose16(self.calc_delta(values)), C.byref(out)),
            "gp_vgicp_factor_compute_error",
        )
        return out.value

    def linearize(self, values):
        """integrated_vgicp_factor_gpu.cpp:185-216"""
        self.linearized = True
        self.linearization_point = self.calc_delta(values)
        if self.linearization_result is not None:
            l = self.linearization_result
            self.linearization_result = None
        else:
            print("warning: performing linearization in sync mode seriously affects the processing speed!!", file=sys.stderr)
            self.touch_points()
            rec = _capi.Linearized6()
            _capi.check(self._lib.gp_vgicp_factor_linearize(self._h, _pose16(self.linearization_point), C.byref(rec)), "gp_vgicp_factor_linearize")
            l = LinearizedSystem6(rec)
            self._num_inliers = l.num_inliers
        if self.is_binary:
            return HessianFactor(self._keys, {(0, 0): l.H_target, (0, 1): l.H_target_source, (1, 1): l.H_source}, [-l.b_target, -l.b_source], l.error)
        return HessianFactor(self._keys, {(0, 0): l.H_source}, [-l.b_source], l.error)

    # ---- NonlinearFactorGPU ----
    def linearization_input_size(self):
        return int(self._lib.gp_vgicp_linearization_input_size())

    def linearization_output_size(self):
        return int(self._lib.gp_vgicp_linearization_output_size())

    def evaluation_input_size(self):
        return int(self._lib.gp_vgicp_evaluation_input_size())

    def evaluation_output_size(self):
        return int(self._lib.gp_vgicp_evaluation_output_size())

    def set_linearization_point(self, values, lin_input_cpu):
        self.touch_points()  # reset_inliers -> touch_points upstream (integrated_vgicp_derivatives_inliers.cu:47)
        lin_input_cpu[:] = np.ascontiguousarray(self.calc_delta(values).T).reshape(16)

    def set_evaluation_point(self, values, eval_input_cpu):
        eval_input_cpu[:] = np.ascontiguousarray(self.calc_delta(values).T).reshape(16)

    def issue_linearize(self, lin_input_cpu, lin_input_gpu, lin_output_gpu):
        _capi.check(self._lib.gp_vgicp_factor_issue_linearize(self._h, lin_input_cpu, lin_input_gpu, lin_output_gpu), "gp_vgicp_factor_issue_linearize")

    def store_linearized(self, lin_output_cpu):
        """integrated_vgicp_factor_gpu.cpp:239-245"""
        self.linearization_result = LinearizedSystem6.from_doubles(lin_output_cpu)
        self.evaluation_result = self.linearization_result.error
        self._num_inliers = self.linearization_result.num_inliers

    def issue_compute_error(self, lin_input_cpu, eval_input_cpu, lin_input_gpu, eval_input_gpu, eval_output_gpu):
        _capi.check(
            self._lib.gp_vgicp_factor_issue_compute_error(self._h, lin_input_cpu, eval_input_cpu, lin_input_gpu, eval_input_gpu, eval_output_gpu),
            "gp_vgicp_factor_issue_compute_error",
        )

    def store_computed_error(self, eval_output_cpu):
        self.evaluation_result = float(eval_output_cpu[0])

    def sync(self):
        _capi.check(self._lib.gp_vgicp_factor_sync(self._h), "gp_vgicp_factor_sync")


class NonlinearFactorSetGPU:
    """NonlinearFactorSetGPU (cuda/nonlinear_factor_set_gpu.cpp:30-228).

    When every registered factor is an IntegratedVGICPFactorGPU the set takes the fast path: one batched
    C-ABI call per linearize()/error() (one H2D, one tiled kernel + one finalize kernel, one D2H).
    Other NonlinearFactorGPU subclasses go through the generic per-factor protocol with the same staging
    buffers and byte cursors as the reference.
    """

    def __init__(self, device="cuda:0"):
        self._lib = _capi.load()
        self.device = device
        self.factors = []
        self.num_linearizations = 0
        self.num_evaluations = 0
        self._batch = None
        self._batch_factors = None
        self._lin_poses = None
        s = C.c_void_p()
        _capi.check(self._lib.gp_stream_create(C.byref(s)), "gp_stream_create")
        self.stream = s

    def __del__(self):
        self._drop_batch()
        if getattr(self, "stream", None):
            self._lib.gp_stream_destroy(self.stream)
            self.stream = None

    def _drop_batch(self):
        if getattr(self, "_batch", None):
            self._lib.gp_vgicp_batch_destroy(self._batch)
            self._batch = None

    def size(self):
        return len(self.factors)

    def clear(self):
        self.factors = []
        self._drop_batch()

    def clear_counts(self):
        self.num_linearizations = 0
        self.num_evaluations = 0

    def linearization_count(self):
        return self.num_linearizations

    def evaluation_count(self):
        return self.num_evaluations

    def add(self, factor):
        """add(factor) keeps only NonlinearFactorGPU instances (:48-56); add(graph) adds each (:58-62)"""
        if isinstance(factor, (list, tuple)):
            for f in factor:
                self.add(f)
            return None
        if isinstance(factor, NonlinearFactorGPU):
            self.factors.append(factor)
            self._drop_batch()
            return True
        return False

    # ---- fast path ----
    def _all_vgicp(self):
        return len(self.factors) > 0 and all(isinstance(f, IntegratedVGICPFactorGPU) for f in self.factors)

    def _ensure_batch(self):
        if self._batch is None:
            arr = (C.c_void_p * len(self.factors))(*[f._h.value for f in self.factors])
            h = C.c_void_p()
            _capi.check(self._lib.gp_vgicp_batch_create(arr, len(self.factors), self.stream, C.byref(h)), "gp_vgicp_batch_create")
            self._batch = h
        return self._batch

    def linearize(self, values):
        if not self.factors:
            return
        self.num_linearizations += self.size()
        if self._all_vgicp():
            F = len(self.factors)
            poses = np.zeros((F, 16))
            for i, f in enumerate(self.factors):
                f.set_linearization_point(values, poses[i])
            view = C.c_void_p()
            _capi.check(self._lib.gp_vgicp_batch_linearize_view(self._ensure_batch(), poses.ctypes.data, C.byref(view)), "gp_vgicp_batch_linearize_view")
            # the records where the finalize kernel stored them (valid until the next call on the batch): consumed without a copy
            out = np.ctypeslib.as_array(C.cast(view, C.POINTER(C.c_double)), shape=(F, _capi.LINEARIZED6_DOUBLES))
            self._lin_poses = poses
            for i, f in enumerate(self.factors):
                f.store_linearized(out[i])
            return
        self._generic_linearize(values)

    def error(self, values):
        if not self.factors:
            return
        self.num_evaluations += self.size()
        if self._all_vgicp():
            F = len(self.factors)
            if self._lin_poses is None or len(self._lin_poses) != F:
                # the reference reuses the last linearize() input buffer (:180-181): error() is only valid after linearize()
                raise _capi.GPError("NonlinearFactorSetGPU.error() called before linearize()")
            poses = np.zeros((F, 16))
            for i, f in enumerate(self.factors):
                f.set_evaluation_point(values, poses[i])
            out = np.zeros(F)
            _capi.check(
                self._lib.gp_vgicp_batch_compute_error(self._ensure_batch(), self._lin_poses.ctypes.data, poses.ctypes.data, out.ctypes.data),
                "gp_vgicp_batch_compute_error",
            )
            for i, f in enumerate(self.factors):
                f.store_computed_error(out[i : i + 1])
            return
        self._generic_error(values)

    def calc_linear_factors(self, linearization_point):
        """:220-228"""
        self.linearize(linearization_point)
        return [f.linearize(linearization_point) for f in self.factors]

    # ---- generic per-factor protocol (cuda/nonlinear_factor_set_gpu.cpp:64-139, 141-218) ----
    def _generic_linearize(self, values):
        import torch

        in_sizes = [f.linearization_input_size() for f in self.factors]
        out_sizes = [f.linearization_output_size() for f in self.factors]
        in_cpu = np.zeros(sum(in_sizes), dtype=np.uint8)
        out_gpu = torch.zeros(sum(out_sizes), dtype=torch.uint8, device=self.device)
        cur = 0
        for f, n in zip(self.factors, in_sizes):
            f.set_linearization_point(values, in_cpu[cur : cur + n].view(np.float64))
            cur += n
        in_gpu = torch.from_numpy(in_cpu).to(self.device)
        torch.cuda.current_stream().synchronize()
        ci = co = 0
        for f, ni, no in zip(self.factors, in_sizes, out_sizes):
            f.issue_linearize(in_cpu.ctypes.data + ci, in_gpu.data_ptr() + ci, out_gpu.data_ptr() + co)
            ci += ni
            co += no
        for f in self.factors:
            f.sync()
        out_cpu = out_gpu.cpu().numpy()
        self._generic_lin_in = (in_cpu, in_gpu, in_sizes)
        co = 0
        for f, no in zip(self.factors, out_sizes):
            f.store_linearized(out_cpu[co : co + no].view(np.float64))
            co += no

    def _generic_error(self, values):
        import torch

        lin_cpu, lin_gpu, lin_sizes = self._generic_lin_in
        in_sizes = [f.evaluation_input_size() for f in self.factors]
        out_sizes = [f.evaluation_output_size() for f in self.factors]
        in_cpu = np.zeros(sum(in_sizes), dtype=np.uint8)
        out_gpu = torch.zeros(sum(out_sizes), dtype=torch.uint8, device=self.device)
        cur = 0
        for f, n in zip(self.factors, in_sizes):
            f.set_evaluation_point(values, in_cpu[cur : cur + n].view(np.float64))
            cur += n
        in_gpu = torch.from_numpy(in_cpu).to(self.device)
        torch.cuda.current_stream().synchronize()
        cl = ci = co = 0
        for f, nl, ni, no in zip(self.factors, lin_sizes, in_sizes, out_sizes):
            f.issue_compute_error(lin_cpu.ctypes.data + cl, in_cpu.ctypes.data + ci, lin_gpu.data_ptr() + cl, in_gpu.data_ptr() + ci, out_gpu.data_ptr() + co)
            cl += nl
            ci += ni
            co += no
        for f in self.factors:
            f.sync()
        out_cpu = out_gpu.cpu().numpy()
        co = 0
        for f, no in zip(self.factors, out_sizes):
            f.store_computed_error(out_cpu[co : co + no].view(np.float64))
            co += no


def create_nonlinear_factor_set_gpu():
    """cuda/nonlinear_factor_set_gpu_create.hpp:10"""
    return NonlinearFactorSetGPU()


class LinearizationHook:
    """optimizers/linearization_hook.{hpp,cpp}: a static list of NonlinearFactorSet factories; every hook
    instance builds one set per registered factory and fans calls out to them."""

    hook_constructors = []

    @staticmethod
    def register_hook(hook):
        LinearizationHook.hook_constructors.append(hook)

    def __init__(self, factors=None):
        self.hooks = [ctor() for ctor in LinearizationHook.hook_constructors]
        if factors is not None:
            self.add(factors)

    def size(self):
        return sum(h.size() for h in self.hooks)

    def clear(self):
        for h in self.hooks:
            h.clear()

    def clear_counts(self):
        for h in self.hooks:
            h.clear_counts()

    def linearization_count(self):
        return sum(h.linearization_count() for h in self.hooks)

    def evaluation_count(self):
        return sum(h.evaluation_count() for h in self.hooks)

    def add(self, factor):
        if isinstance(factor, (list, tuple)):
            for f in factor:
                self.add(f)
            return None
        return any(h.add(factor) for h in self.hooks)

    def linearize(self, values):
        for h in self.hooks:
            h.linearize(values)

    def error(self, values):
        for h in self.hooks:
            h.error(values)

    def calc_linear_factors(self, values):
        out = []
        for h in self.hooks:
            out.extend(h.calc_linear_factors(values))
        return out
