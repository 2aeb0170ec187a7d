"""bench_lm.py -- measurement harness (NOT product code): what ONE Levenberg-Marquardt iteration over a graph of VGICP factors costs host to host, split by phase.

The reference's only timing harness times the whole optimize() loop (src/demo/demo_benchmark.cpp:98-230, gate :220-229).  This module restates the CADENCE of
LevenbergMarquardtOptimizerExt (src/gtsam_points/optimizers/levenberg_marquardt_ext.cpp) -- nothing of its bookkeeping:
  iterate()   :352-392   linearize(values)  ->  error(values) (free: the GPU factor keeps the linearise's error, integrated_vgicp_factor_gpu.cpp:239-245)
  tryLambda() :188-350   buildDampedSystem(lambda) -> solve -> retract -> error(new values), evaluated on the correspondences / fused covariances frozen at the
                         linearisation point -> accept (lambda /= 10) when the nonlinear cost dropped and costChange / linearizedCostChange > minModelFidelity (1e-3),
                         else lambda *= 10 and try again; GTSAM defaults lambdaInitial 1e-5, lambdaFactor 10, lambdaUpperBound 1e5, relativeErrorTol 1e-5
  optimize()  :394-430   until |delta error| is below the tolerances or maxIterations
over two interchangeable back ends:
  GpuGraph  one gp_vgicp_batch (ONE batched launch per linearise / error evaluation); records stay in HBM and feed the block-sparse LL^T of gp_sparse.hip,
            or -- solver="host" -- go to the host for a numpy solve
  CpuGraph  the checker's CPU factors (oracle/_ref/libref.so = the reference's own IntegratedVGICPFactor, else the C restatement), sequential over factors as
            graph_.linearize does, numpy solve: the cpu_baseline of the same loop
Poses: values[k] = 4x4; pose `fixed` is held (the demo pins it with a 1e6-precision prior).  The update is the right-multiplicative retract of gtsam::Pose3 in the
(omega, v) order of the factor's Jacobians.
"""
import ctypes as C
import time

import numpy as np


def expmap_many(xi):
    """Pose3::Expmap for [n, 6] twists (omega, v) -> [n, 4, 4]"""
    xi = np.asarray(xi, dtype=np.float64).reshape(-1, 6)
    n = len(xi)
    w, v = xi[:, :3], xi[:, 3:]
    th2 = np.einsum("ij,ij->i", w, w)
    th = np.sqrt(th2)
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    A = np.where(small, 1.0 - th2 / 6.0, np.sin(ths) / ths)
    B = np.where(small, 0.5 - th2 / 24.0, (1.0 - np.cos(ths)) / (ths * ths))
    Cc = np.where(small, 1.0 / 6.0 - th2 / 120.0, (ths - np.sin(ths)) / (ths * ths * ths))
    W = np.zeros((n, 3, 3))
    W[:, 0, 1], W[:, 0, 2], W[:, 1, 0], W[:, 1, 2], W[:, 2, 0], W[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    W2 = W @ W
    I = np.eye(3)[None]
    R = I + A[:, None, None] * W + B[:, None, None] * W2
    V = I + B[:, None, None] * W + Cc[:, None, None] * W2
    T = np.zeros((n, 4, 4))
    T[:, :3, :3] = R
    T[:, :3, 3] = np.einsum("nij,nj->ni", V, v)
    T[:, 3, 3] = 1.0
    return T


def inv_many(T):
    R, t = T[:, :3, :3], T[:, :3, 3]
    out = np.zeros_like(T)
    Rt = R.transpose(0, 2, 1)
    out[:, :3, :3] = Rt
    out[:, :3, 3] = -np.einsum("nij,nj->ni", Rt, t)
    out[:, 3, 3] = 1.0
    return out


def pose_error(T, G):
    """(rotation angle, translation norm) of G^-1 T"""
    E = np.linalg.inv(G) @ T
    c = np.clip((np.trace(E[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)
    return float(np.arccos(c)), float(np.linalg.norm(E[:3, 3]))


class _Graph:
    """key bookkeeping shared by the back ends: pairs [(target key, source key)], keys 0..N-1, one of them fixed"""

    def __init__(self, pairs, num_poses, fixed=0):
        self.pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        self.N = int(num_poses)
        self.fixed = int(fixed)
        slot = np.full(self.N, -1, dtype=np.int64)
        k = 0
        for i in range(self.N):
            if i != self.fixed:
                slot[i] = k
                k += 1
        self.slot = slot
        self.num_slots = k
        self.factor_slots = np.stack([slot[self.pairs[:, 0]], slot[self.pairs[:, 1]]], axis=1).astype(np.int32)

    def deltas(self, values):
        return inv_many(values[self.pairs[:, 0]]) @ values[self.pairs[:, 1]]

    def retract(self, values, dx):
        out = values.copy()
        idx = np.nonzero(self.slot >= 0)[0]
        out[idx] = values[idx] @ expmap_many(dx.reshape(-1, 6)[self.slot[idx]])
        return out


def _poses16(deltas):
    return np.ascontiguousarray(deltas.transpose(0, 2, 1)).reshape(len(deltas), 16)  # column-major 4x4 per factor


def host_system(rec, factor_slots, num_slots):
    """DenseLinearSystemBuilder over gp_linearized6 records on the host (checker / CPU back end): A, b = sum of -b_t / -b_s, c = sum of errors"""
    n = 6 * num_slots
    A, b = np.zeros((n, n)), np.zeros(n)
    for k, (st, ss) in enumerate(factor_slots):
        Ht, Hs, Hts = rec[k, 2:38].reshape(6, 6).T, rec[k, 38:74].reshape(6, 6).T, rec[k, 74:110].reshape(6, 6).T
        bt, bs = rec[k, 110:116], rec[k, 116:122]
        if st >= 0:
            A[6 * st : 6 * st + 6, 6 * st : 6 * st + 6] += Ht
            b[6 * st : 6 * st + 6] -= bt
        if ss >= 0:
            A[6 * ss : 6 * ss + 6, 6 * ss : 6 * ss + 6] += Hs
            b[6 * ss : 6 * ss + 6] -= bs
        if st >= 0 and ss >= 0:
            A[6 * st : 6 * st + 6, 6 * ss : 6 * ss + 6] += Hts
            A[6 * ss : 6 * ss + 6, 6 * st : 6 * st + 6] += Hts.T
    return A, b, float(rec[:, 1].sum())


class GpuGraph(_Graph):
    """factors: IntegratedVGICPFactorGPU objects in `pairs` order, all on one device"""

    name = "gpu"

    def __init__(self, gpa, factors, pairs, num_poses, fixed=0, solver="device", stream=None, device="cuda:0"):
        import torch

        super().__init__(pairs, num_poses, fixed)
        from gtsam_points_amd import _capi

        self._capi, self._lib, self._torch = _capi, gpa.load(), torch
        self.factors = list(factors)
        F = len(self.factors)
        arr = (C.c_void_p * F)(*[f._h.value for f in self.factors])
        self.batch = C.c_void_p()
        _capi.check(self._lib.gp_vgicp_batch_create(arr, F, stream, C.byref(self.batch)), "gp_vgicp_batch_create")
        self.solver = solver
        self.rec_dev = torch.zeros((F, _capi.LINEARIZED6_DOUBLES), dtype=torch.float64, device=device)
        self.rec_ptr = C.c_void_p(self.rec_dev.data_ptr())
        self.errs = np.zeros(F)
        self.rec_host = np.zeros((F, _capi.LINEARIZED6_DOUBLES))
        if solver in ("device", "device-three-calls"):
            self.sys = gpa.SparseLinearSystemGPU(self.num_slots, self.factor_slots, ordering="auto", stream=stream) if self.num_slots > 1 else gpa.DenseLinearSystemGPU(
                self.num_slots, self.factor_slots, stream=stream)
            self._download = self._lib.gp_sparse_system_download if self.num_slots > 1 else self._lib.gp_dense_system_download
        self.b = np.zeros(6 * self.num_slots)
        self.c = np.zeros(1)
        self.x = np.zeros(6 * self.num_slots)
        self.poses_lin = None
        self.sync_phases = False  # True: wait for the linearise before the solve is issued, so that the phase split is the device's, not the queue's
        torch.cuda.synchronize()

    def close(self):
        if self.batch:
            self._lib.gp_vgicp_batch_destroy(self.batch)
            self.batch = None

    def linearize(self, values):
        """-> total error at `values`; the system (records) stays where the solver wants it"""
        self.poses_lin = _poses16(self.deltas(values))
        if self.solver != "host":
            self._capi.check(self._lib.gp_vgicp_batch_issue_linearize(self.batch, self.poses_lin.ctypes.data, self.rec_ptr), "gp_vgicp_batch_issue_linearize")
            # (no synchronisation: the solver's build is ordered behind it on the same stream; the error comes back with b, below)
            if self.sync_phases:
                self._capi.check(self._lib.gp_vgicp_batch_sync(self.batch), "gp_vgicp_batch_sync")
            return None
        self._capi.check(self._lib.gp_vgicp_batch_linearize(self.batch, self.poses_lin.ctypes.data, self.rec_host.ctypes.data), "gp_vgicp_batch_linearize")
        self.A, self.b, c = host_system(self.rec_host, self.factor_slots, self.num_slots)
        return c

    def solve(self, lam):
        """-> (dx, b, error at the linearisation point)"""
        if self.solver == "device":  # buildDampedSystem + solve as ONE call with one wait (gp_sparse_system_step / gp_dense_system_step)
            return self.sys.step(self.rec_dev, lam=lam, out=(self.x, self.b, self.c))
        if self.solver == "device-three-calls":  # (round 4's form, kept for the A/B: build, download of b and c, solve -- two waits, four copies)
            self.sys.build(self.rec_dev, lam=lam)
            self._capi.check(self._download(self.sys._h, None, self.b.ctypes.data, self.c.ctypes.data), "system_download")
            return self.sys.solve(), self.b, float(self.c[0])
        n = len(self.b)
        return np.linalg.solve(self.A + lam * np.eye(n), self.b), self.b, None

    def error(self, values):
        pe = _poses16(self.deltas(values))
        self._capi.check(self._lib.gp_vgicp_batch_compute_error(self.batch, self.poses_lin.ctypes.data, pe.ctypes.data, self.errs.ctypes.data), "gp_vgicp_batch_compute_error")
        return float(self.errs.sum())


class GpuTrialGraph(_Graph):
    """round 6: the values live in device memory (gp_lm_graph_*, csrc/gp_lm.hip) -- linearize() issues the batch's linearise at them, solve() is the whole trial (damped
    step + retract + error evaluation at the trial values, ONE wait), retract() / error() hand back what the trial already computed.  Same cadence, same tests, driven by
    the same run_lm; `native_loop` runs the library's own loop (gp_lm_graph_optimize) instead."""

    name = "gpu-trial"

    def __init__(self, gpa, factors, pairs, num_poses, fixed=0, stream=None):
        super().__init__(pairs, num_poses, fixed)
        self.g = gpa.LevenbergMarquardtGraphGPU(factors, self.pairs, num_poses, fixed=(fixed,), stream=stream)
        self.sync_phases = False
        self._trial = None
        self._trial_error = None

    def close(self):
        self.g.close()

    def linearize(self, values):
        if values is self._trial and self._trial is not None:
            self.g.accept()
        else:
            self.g.set_values(values)
        self._trial = None
        self.g.linearize()
        if self.sync_phases:
            self.g.sync()
        return None

    def solve(self, lam):
        dx, b, c, e, v = self.g.try_lambda(lam, want_values=True)
        self._trial, self._trial_error = v, e
        return dx, b, c

    def retract(self, values, dx):
        return self._trial

    def error(self, values):
        assert values is self._trial
        return self._trial_error

    def native_loop(self, values0, max_iterations=30):
        """-> run_lm's dict from ONE call of gp_lm_graph_optimize (no per-phase split: the loop never returns to the interpreter)"""
        self.g.set_values(values0)
        self.g.sync()
        t0 = time.perf_counter()
        values, s = self.g.optimize(max_iterations=max_iterations)
        total = time.perf_counter() - t0
        return dict(values=values, iterations=s["iterations"], inner_iterations=s["inner_iterations"], errors=[], steps=[], seconds=total,
                    phases=dict(linearize=0.0, solve=total, error=0.0, glue=0.0), final_error=s["final_error"], final_lambda=s["final_lambda"])


class CpuGraph(_Graph):
    """cpu_factors[k]: linearize(delta) -> Linearized6 and error(delta) (= evaluate on the state of the last linearize, integrated_matching_cost_factor.cpp:32-35); sequential over factors"""

    name = "cpu"

    def __init__(self, cpu_factors, pairs, num_poses, fixed=0):
        super().__init__(pairs, num_poses, fixed)
        self.f = list(cpu_factors)
        self.rec = np.zeros((len(self.f), 122))

    def close(self):
        pass

    def linearize(self, values):
        self.d_lin = self.deltas(values)
        for k, f in enumerate(self.f):
            L = f.linearize(self.d_lin[k])
            self.rec[k] = np.concatenate([[L.num_inliers, L.error], L.H_target.T.ravel(), L.H_source.T.ravel(), L.H_target_source.T.ravel(), L.b_target, L.b_source])
        self.A, self.b, c = host_system(self.rec, self.factor_slots, self.num_slots)
        return c

    def solve(self, lam):
        return np.linalg.solve(self.A + lam * np.eye(len(self.b)), self.b), self.b, None

    def error(self, values):
        d = self.deltas(values)
        return float(sum(f.error(d[k]) for k, f in enumerate(self.f)))  # on the correspondences / fused covariances of the last linearize


def run_lm(graph, values0, max_iterations=20, lambda0=1e-5, factor=10.0, lambda_max=1e5, rel_tol=1e-5, abs_tol=1e-5, min_fidelity=1e-3, time_budget_s=None):
    """-> dict(values, iterations, inner_iterations, errors, per-phase seconds)"""
    values = np.asarray(values0, dtype=np.float64).copy()
    lam = lambda0
    t = dict(linearize=0.0, solve=0.0, error=0.0, glue=0.0)
    errors, steps, inner, iters = [], [], 0, 0
    t_start = time.perf_counter()
    err = None
    for it in range(max_iterations):
        iters += 1
        t0 = time.perf_counter()
        e_lin = graph.linearize(values)
        t["linearize"] += time.perf_counter() - t0
        if e_lin is not None:
            err = e_lin
        stop = False
        while True:
            inner += 1
            t0 = time.perf_counter()
            try:
                dx, b, e0 = graph.solve(lam)
                ok = True
            except Exception:  # GP_ERROR_INDETERMINATE / LinAlgError: IndeterminantLinearSystemException upstream
                ok = False
            t["solve"] += time.perf_counter() - t0
            if ok and e0 is not None:
                err = e0
            accepted = False
            if ok:
                t0 = time.perf_counter()
                lin_change = 0.5 * float(b @ dx) + 0.5 * lam * float(dx @ dx)  # old - new linearised error of (A + lam I) dx = b
                new_values = graph.retract(values, dx)
                t["glue"] += time.perf_counter() - t0
                if lin_change >= 0.0:
                    t0 = time.perf_counter()
                    new_err = graph.error(new_values)
                    t["error"] += time.perf_counter() - t0
                    change = err - new_err
                    accepted = lin_change > np.finfo(float).eps * err and change / lin_change > min_fidelity
                    if abs(change) < rel_tol * err:
                        stop = True
            if accepted:
                prev = err
                steps.append((prev, new_err))  # (cost at the linearisation point, cost of the accepted step on the same correspondences)
                values, err = new_values, new_err
                lam = lam / factor
                errors.append(err)
                if abs(prev - err) < abs_tol or abs(prev - err) / max(prev, 1e-300) < rel_tol:
                    stop = True
                break
            if stop:
                break
            lam *= factor
            if lam >= lambda_max:
                stop = True
                break
        if stop:
            break
        if time_budget_s is not None and time.perf_counter() - t_start > time_budget_s:
            break
    total = time.perf_counter() - t_start
    t["glue"] += total - sum(t.values())
    return dict(values=values, iterations=iters, inner_iterations=inner, errors=errors, steps=steps, seconds=total, phases=t, final_error=err, final_lambda=lam)


def summarize(res, graph, truth, label):
    """the bench object of one run: per-iteration ms by phase, iterations to the reference's alignment gate (test_matching_cost_factors.cpp:227: 0.015 rad / 0.15 m)"""
    ang = tr = 0.0
    base_v, base_t = np.linalg.inv(res["values"][graph.fixed]), np.linalg.inv(truth[graph.fixed])
    for k in range(graph.N):
        a, d = pose_error(base_v @ res["values"][k], base_t @ truth[k])
        ang, tr = max(ang, a), max(tr, d)
    it = max(res["iterations"], 1)
    ph = {k: round(v / it * 1e3, 4) for k, v in res["phases"].items()}
    return dict(backend=label, iterations=res["iterations"], inner_iterations=res["inner_iterations"], ms_total=round(res["seconds"] * 1e3, 3), ms_per_iteration=round(res["seconds"] / it * 1e3, 4),
                ms_per_iteration_by_phase=ph, dominant_phase=max(ph, key=ph.get), final_error=res["final_error"], max_rotation_error_rad=round(ang, 5), max_translation_error_m=round(tr, 5),
                gate_met=bool(ang < 0.015 and tr < 0.15))
