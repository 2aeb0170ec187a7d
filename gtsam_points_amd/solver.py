"""The step after the path: the damped normal equations of a graph of VGICP factors, assembled and solved on the device.

  DenseLinearSystemGPU  <- DenseLinearSystemBuilder (optimizers/linear_system_builder.cpp:39-48)
                           + buildDampedSystem       (optimizers/levenberg_marquardt_ext.cpp:146-161)
                           + DenseLinearSolver::solve (optimizers/linear_solver.hpp:18-22)
  SparseLinearSystemGPU <- SparseLinearSystemBuilder<6> (optimizers/linear_system_builder.hpp:41-72) + buildDampedSystem
                           + SparseLinearSolver::solve (optimizers/linear_solver.hpp:24-29): block-sparse LL^T over the pose graph
  linearize_on_device   -- one batched linearise whose records stay in HBM (what the solver consumes)

Poses that are variables get a slot (0..num_slots-1); a factor key without a slot (a fixed pose) drops out of the system.
"""
import ctypes as C

import numpy as np

from . import _capi


def linearize_on_device(factors, values, device="cuda:0", stream=None):
    """Batched linearise of IntegratedVGICPFactorGPU objects; returns a [F, 122] float64 CUDA tensor of gp_linearized6
    records (nothing is copied to the host)."""
    import torch

    lib = _capi.load()
    F = len(factors)
    poses = np.zeros((F, 16))
    for i, f in enumerate(factors):
        f.set_linearization_point(values, poses[i])
    arr = (C.c_void_p * F)(*[f._h.value for f in factors])
    batch = C.c_void_p()
    _capi.check(lib.gp_vgicp_batch_create(arr, F, stream, C.byref(batch)), "gp_vgicp_batch_create")
    out = torch.zeros((F, _capi.LINEARIZED6_DOUBLES), dtype=torch.float64, device=device)
    torch.cuda.current_stream(out.device).synchronize()
    try:
        _capi.check(lib.gp_vgicp_batch_issue_linearize(batch, poses.ctypes.data, C.c_void_p(out.data_ptr())), "gp_vgicp_batch_issue_linearize")
        _capi.check(lib.gp_vgicp_batch_sync(batch), "gp_vgicp_batch_sync")
    finally:
        lib.gp_vgicp_batch_destroy(batch)
    return out


class DenseLinearSystemGPU:
    """A x = b over 6-dof pose slots, built from stacked device-resident gp_linearized6 records.

    factor_slots: [(target_slot, source_slot)] per factor; a negative slot = that pose is not a variable."""

    def __init__(self, num_slots, factor_slots, stream=None):
        self._lib = _capi.load()
        self.num_slots = int(num_slots)
        self.factor_slots = np.ascontiguousarray(np.asarray(factor_slots, dtype=np.int32).reshape(-1, 2))
        self.stream = stream
        h = C.c_void_p()
        _capi.check(self._lib.gp_dense_system_create(self.num_slots, self.factor_slots.ctypes.data, len(self.factor_slots), stream, C.byref(h)), "gp_dense_system_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.gp_dense_system_destroy(h)
            self._h = None

    @property
    def size(self):
        return 6 * self.num_slots

    def build(self, records_dev, lam=0.0, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32, prior_diag=None):
        """records_dev: [F, 122] float64 CUDA tensor.  lam / diagonal_damping: buildDampedSystem; prior_diag: optional
        extra diagonal (length 6 * num_slots)."""
        if tuple(records_dev.shape) != (len(self.factor_slots), _capi.LINEARIZED6_DOUBLES) or not records_dev.is_contiguous():
            raise ValueError("records_dev must be a contiguous [num_factors, 122] float64 device tensor")
        prior = None
        if prior_diag is not None:
            prior = np.ascontiguousarray(prior_diag, dtype=np.float64)
            if prior.shape != (self.size,):
                raise ValueError("prior_diag must have 6 * num_slots entries")
        _capi.check(
            self._lib.gp_dense_system_build(self._h, C.c_void_p(records_dev.data_ptr()), float(lam), int(bool(diagonal_damping)), float(min_diagonal), float(max_diagonal),
                                            prior.ctypes.data if prior is not None else None),
            "gp_dense_system_build",
        )
        return self

    def download(self):
        n = self.size
        A, b, c = np.zeros((n, n)), np.zeros(n), np.zeros(1)
        _capi.check(self._lib.gp_dense_system_download(self._h, A.ctypes.data, b.ctypes.data, c.ctypes.data), "gp_dense_system_download")
        return A.T.copy(), b, float(c[0])  # column-major -> numpy (symmetric anyway)

    def solve(self):
        """x with A x = b (consumes the built system); raises GPError when A is not positive definite."""
        x = np.zeros(self.size)
        _capi.check(self._lib.gp_dense_system_solve(self._h, x.ctypes.data, None), "gp_dense_system_solve")
        return x

    def set_one_launch(self, enable=True):
        """a system of ONE pose runs step() as one launch (default); False: the multi-launch path (bit-identical: tests, A/B).  -> what the next step() runs"""
        return bool(self._lib.gp_dense_system_set_one_launch(self._h, 1 if enable else 0))

    def step(self, records_dev, lam=0.0, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32, prior_diag=None, out=None):
        """build + download(b, c) + solve as ONE call with one synchronisation -- two with a prior_diag -- (gp_dense_system_step): -> (x, b, c); the optimizer's tryLambda
        (levenberg_marquardt_ext.cpp:188-260).  out: optional (x, b, c) float64 arrays to fill ([n], [n], [1]) instead of new ones.  Raises GPError (indeterminate) when the
        damped system is not positive definite; out's b and c are filled even then."""
        if tuple(records_dev.shape) != (len(self.factor_slots), _capi.LINEARIZED6_DOUBLES) or not records_dev.is_contiguous():
            raise ValueError("records_dev must be a contiguous [num_factors, 122] float64 device tensor")
        prior = None
        if prior_diag is not None:
            prior = np.ascontiguousarray(prior_diag, dtype=np.float64)
            if prior.shape != (self.size,):
                raise ValueError("prior_diag must have 6 * num_slots entries")
        x, b, c = _step_out(out, self.size)
        _capi.check(
            self._lib.gp_dense_system_step(self._h, C.c_void_p(records_dev.data_ptr()), float(lam), int(bool(diagonal_damping)), float(min_diagonal), float(max_diagonal),
                                            prior.ctypes.data if prior is not None else None, x.ctypes.data, b.ctypes.data, c.ctypes.data),
            "gp_dense_system_step",
        )
        return x, b, float(c[0])


def _step_out(out, n):
    """(x, b, c) buffers of a step() call: the C entry points memcpy n, n and 1 doubles into them through their raw pointers, so anything that is not a
    C-contiguous float64 array of exactly that shape is refused here"""
    if out is None:
        return np.zeros(n), np.zeros(n), np.zeros(1)
    if len(out) != 3:
        raise ValueError("out must be (x, b, c)")
    for arr, shape, name in zip(out, ((n,), (n,), (1,)), "xbc"):
        if not (isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.shape == shape and arr.flags.c_contiguous and arr.flags.writeable):
            raise ValueError(f"out[{name}] must be a writable C-contiguous float64 array of shape {shape}")
    return out


def sparse_symbolic(num_slots, factor_slots, ordering=0):
    """The symbolic phase of SparseLinearSystemGPU alone (host code, no device needed): dict(perm, parent, nnz_a_blocks,
    nnz_l_blocks, num_subtrees, top_columns)."""
    lib = _capi.load()
    fs = np.ascontiguousarray(np.asarray(factor_slots, dtype=np.int32).reshape(-1, 2))
    perm, parent = np.zeros(num_slots, np.int32), np.zeros(num_slots, np.int32)
    na, nl, ns, nt = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
    _capi.check(lib.gp_sparse_symbolic(int(num_slots), fs.ctypes.data, len(fs), int(ordering), perm.ctypes.data, parent.ctypes.data, C.byref(na), C.byref(nl), C.byref(ns),
                                       C.byref(nt)), "gp_sparse_symbolic")
    lv, cc, wl = C.c_int(), C.c_int(), C.c_int()
    _capi.check(lib.gp_sparse_symbolic_schedule(int(num_slots), fs.ctypes.data, len(fs), int(ordering), C.byref(lv), C.byref(cc), C.byref(wl)), "gp_sparse_symbolic_schedule")
    arr = [np.zeros(max(wl.value, 1), np.int32) for _ in range(4)]
    _capi.check(lib.gp_debug_sparse_work_lists(int(num_slots), fs.ctypes.data, len(fs), int(ordering), len(arr[0]), *[a.ctypes.data for a in arr]), "gp_debug_sparse_work_lists")
    lists = [dict(level=int(arr[0][i]), columns=int(arr[1][i]), products=int(arr[2][i]), max_column_products=int(arr[3][i])) for i in range(wl.value)]
    return dict(perm=perm, parent=parent, nnz_a_blocks=na.value, nnz_l_blocks=nl.value, num_subtrees=ns.value, top_columns=nt.value, num_levels=lv.value,
                critical_columns=cc.value, num_lists=wl.value, work_lists=lists)


class SparseLinearSystemGPU:
    """A x = b over 6-dof pose slots as a block-sparse lower triangle, solved by a block-sparse LL^T on the device.

    factor_slots as for DenseLinearSystemGPU; ordering: "natural" (slot order = elimination order), "nd" (nested dissection), "amd" (minimum degree by
    multiple elimination: the fill of the COLAMD-class orderings GTSAM gives the reference), "amd1" (the same with a slack of one on the degree) or
    "auto" (default: "nd" and "amd1" are both tried, the schedule with the shorter critical path is kept)."""

    ORDERINGS = {"natural": 0, "nd": 1, "amd": 2, "amd1": 3, "auto": 4}

    def __init__(self, num_slots, factor_slots, ordering="auto", stream=None):
        self._lib = _capi.load()
        self.num_slots = int(num_slots)
        self.factor_slots = np.ascontiguousarray(np.asarray(factor_slots, dtype=np.int32).reshape(-1, 2))
        self.stream = stream
        h = C.c_void_p()
        _capi.check(self._lib.gp_sparse_system_create(self.num_slots, self.factor_slots.ctypes.data, len(self.factor_slots), self.ORDERINGS[ordering], stream, C.byref(h)),
                    "gp_sparse_system_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.gp_sparse_system_destroy(h)
            self._h = None

    @property
    def size(self):
        return 6 * self.num_slots

    def set_one_launch(self, enable=True):
        """step() of a system whose factor fits one compute unit's LDS (<= 128 poses, <= ~440 blocks of L) runs as ONE launch by default; False selects the multi-launch
        form, "teams" the one-launch step's first form (every list a team of waves in lock step), "lone-waves" its second (a work list per wave; the default gives a list a team of
        waves that meet without workgroup barriers where the level has at most four lists, and a lone wave otherwise) -- all bit-identical: the switch is for the test that
        says so and for timing.  Returns what the next step() runs (True: one launch)."""
        return bool(self._lib.gp_sparse_system_set_one_launch(self._h, 2 if enable == "teams" else (3 if enable == "lone-waves" else (1 if enable else 0))))

    def info(self):
        na, nl, bp, ns, nt = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        _capi.check(self._lib.gp_sparse_system_info(self._h, C.byref(na), C.byref(nl), C.byref(bp), C.byref(ns), C.byref(nt)), "gp_sparse_system_info")
        return dict(nnz_a_blocks=na.value, nnz_l_blocks=nl.value, block_products=bp.value, num_subtrees=ns.value, top_columns=nt.value)

    def build(self, records_dev, lam=0.0, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32, prior_diag=None):
        if tuple(records_dev.shape) != (len(self.factor_slots), _capi.LINEARIZED6_DOUBLES) or not records_dev.is_contiguous():
            raise ValueError("records_dev must be a contiguous [num_factors, 122] float64 device tensor")
        prior = None
        if prior_diag is not None:
            prior = np.ascontiguousarray(prior_diag, dtype=np.float64)
            if prior.shape != (self.size,):
                raise ValueError("prior_diag must have 6 * num_slots entries")
        _capi.check(
            self._lib.gp_sparse_system_build(self._h, C.c_void_p(records_dev.data_ptr()), float(lam), int(bool(diagonal_damping)), float(min_diagonal), float(max_diagonal),
                                             prior.ctypes.data if prior is not None else None),
            "gp_sparse_system_build",
        )
        return self

    def download(self):
        """(A dense symmetric [n, n], b, c) in slot order -- for checkers"""
        n = self.size
        A, b, c = np.zeros((n, n)), np.zeros(n), np.zeros(1)
        _capi.check(self._lib.gp_sparse_system_download(self._h, A.ctypes.data, b.ctypes.data, c.ctypes.data), "gp_sparse_system_download")
        return A.T.copy(), b, float(c[0])

    def solve(self):
        x = np.zeros(self.size)
        _capi.check(self._lib.gp_sparse_system_solve(self._h, x.ctypes.data, None), "gp_sparse_system_solve")
        return x

    def step(self, records_dev, lam=0.0, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32, prior_diag=None, out=None):
        """build + download(b, c) + solve as ONE call with one synchronisation (gp_sparse_system_step): -> (x, b, c); the optimizer's tryLambda
        (levenberg_marquardt_ext.cpp:188-260).  out: optional (x, b, c) float64 arrays to fill ([n], [n], [1]) instead of new ones.  Raises GPError (indeterminate) when the
        damped system is not positive definite; out's b and c are filled even then."""
        if tuple(records_dev.shape) != (len(self.factor_slots), _capi.LINEARIZED6_DOUBLES) or not records_dev.is_contiguous():
            raise ValueError("records_dev must be a contiguous [num_factors, 122] float64 device tensor")
        prior = None
        if prior_diag is not None:
            prior = np.ascontiguousarray(prior_diag, dtype=np.float64)
            if prior.shape != (self.size,):
                raise ValueError("prior_diag must have 6 * num_slots entries")
        x, b, c = _step_out(out, self.size)
        _capi.check(
            self._lib.gp_sparse_system_step(self._h, C.c_void_p(records_dev.data_ptr()), float(lam), int(bool(diagonal_damping)), float(min_diagonal), float(max_diagonal),
                                            prior.ctypes.data if prior is not None else None, x.ctypes.data, b.ctypes.data, c.ctypes.data),
            "gp_sparse_system_step",
        )
        return x, b, float(c[0])


class LevenbergMarquardtGraphGPU:
    """The optimizer's trial with the VALUES in device memory (gp_lm_graph_*, csrc/gp_lm.hip): what LevenbergMarquardtOptimizerExt does with its GPU factor set
    per trial (optimizers/levenberg_marquardt_ext.cpp: linearization_hook_->linearize(values) :352-392, buildDampedSystem + solve + retract + linearization_hook_->error(newValues)
    :188-350) as linearize() / try_lambda() / accept(), one wait per trial; optimize() runs the reference's loop over them natively.

    factors: IntegratedVGICPFactorGPU objects; pairs[i] = (target pose, source pose) of factor i, poses 0..num_poses-1; fixed: indices of held poses.
    values are [num_poses, 4, 4] float64 arrays (rigid)."""

    def __init__(self, factors, pairs, num_poses, fixed=(0,), ordering="auto", stream=None):
        self._lib = _capi.load()
        self.factors = list(factors)  # (kept alive: the batch holds their handles)
        F = len(self.factors)
        self.pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        if len(self.pairs) != F:
            raise ValueError(f"{F} factors, {len(self.pairs)} pose pairs")
        self.num_poses = int(num_poses)
        held = np.zeros(self.num_poses, dtype=np.uint8)
        held[list(fixed)] = 1
        self._batch, self._h = C.c_void_p(), C.c_void_p()
        arr = (C.c_void_p * F)(*[f._h.value for f in self.factors])
        _capi.check(self._lib.gp_vgicp_batch_create(arr, F, stream, C.byref(self._batch)), "gp_vgicp_batch_create")
        try:
            _capi.check(self._lib.gp_lm_graph_create(self._batch, self.pairs.ctypes.data, self.num_poses, held.ctypes.data, SparseLinearSystemGPU.ORDERINGS[ordering], C.byref(self._h)), "gp_lm_graph_create")
        except Exception:
            self._lib.gp_vgicp_batch_destroy(self._batch)
            self._batch = None
            raise
        self.n = self._lib.gp_lm_graph_num_variables(self._h)
        self._x, self._b, self._c, self._e = np.zeros(self.n), np.zeros(self.n), np.zeros(1), np.zeros(1)
        self._v = np.zeros((self.num_poses, 16))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gp_lm_graph_destroy(self._h)
            self._h = None
        if getattr(self, "_batch", None):
            self._lib.gp_vgicp_batch_destroy(self._batch)
            self._batch = None

    __del__ = close

    @staticmethod
    def _to16(values):
        v = np.asarray(values, dtype=np.float64)
        return np.ascontiguousarray(v.transpose(0, 2, 1)).reshape(len(v), 16)

    def set_values(self, values):
        v = self._to16(values)
        if v.shape != (self.num_poses, 16):
            raise ValueError(f"values must be [{self.num_poses}, 4, 4]")
        _capi.check(self._lib.gp_lm_graph_set_values(self._h, v.ctypes.data), "gp_lm_graph_set_values")

    def values(self):
        _capi.check(self._lib.gp_lm_graph_get_values(self._h, self._v.ctypes.data), "gp_lm_graph_get_values")
        return self._v.reshape(self.num_poses, 4, 4).transpose(0, 2, 1).copy()

    def linearize(self):
        """asynchronous: records at the current values stay in HBM"""
        _capi.check(self._lib.gp_lm_graph_linearize(self._h), "gp_lm_graph_linearize")

    def sync(self):
        _capi.check(self._lib.gp_vgicp_batch_sync(self._batch), "gp_vgicp_batch_sync")

    def try_lambda(self, lam, diagonal=False, min_diagonal=1e-6, max_diagonal=1e32, want_values=False):
        """-> (dx, b, cost at the linearisation point, cost at the trial values[, trial values]); GPError (code 5) on an indeterminate system.
        The arrays are the object's own buffers: valid until the next call."""
        _capi.check(self._lib.gp_lm_graph_try_lambda(self._h, float(lam), int(bool(diagonal)), float(min_diagonal), float(max_diagonal), self._x.ctypes.data, self._b.ctypes.data,
                                                     self._c.ctypes.data, self._e.ctypes.data, self._v.ctypes.data if want_values else None), "gp_lm_graph_try_lambda")
        out = (self._x, self._b, float(self._c[0]), float(self._e[0]))
        return out + (self._v.reshape(self.num_poses, 4, 4).transpose(0, 2, 1).copy(),) if want_values else out

    def accept(self):
        _capi.check(self._lib.gp_lm_graph_accept(self._h), "gp_lm_graph_accept")

    def set_one_launch(self, enable=True):
        """the graph's own damped system: False = its multi-launch step with the retract as a kernel of its own behind it (same bits) -> what the next trial runs"""
        return int(self._lib.gp_lm_graph_set_one_launch(self._h, int(bool(enable))))

    def set_speculation(self, enable):
        """queue the linearise at the trial values behind each trial (default on; same results either way) -> previous setting"""
        return bool(self._lib.gp_lm_graph_set_speculation(self._h, int(bool(enable))))

    def optimize(self, values=None, **params):
        """the reference's loop, natively; params: fields of gp_lm_params (GTSAM's defaults otherwise).  -> (values, summary dict)"""
        if values is not None:
            self.set_values(values)
        p, s = _capi.LmParams(), _capi.LmSummary()
        self._lib.gp_lm_params_default(C.byref(p))
        for k, v in params.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown parameter {k}")
            setattr(p, k, v)
        _capi.check(self._lib.gp_lm_graph_optimize(self._h, C.byref(p), C.byref(s)), "gp_lm_graph_optimize")
        return self.values(), {k: getattr(s, k) for k, _ in s._fields_ if k != "reserved_"}
