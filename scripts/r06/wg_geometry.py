"""Round 6: the C2 headline step as bench.py runs it (synchronous fused call) with 4- (product), 8- and 16-wave workgroups (GP_TUNE_WG_WAVES), alternating:
the kernel's own 100 MHz stamps inside the steps (streaming part, whole fused kernel), host wall per step; per geometry also fused == two-kernel bit for bit, the error
evaluation against the 4-wave form, and the relative difference of the record to the 4-wave record (different partition: ~1e-16).  One JSON object per line.
Run under `rocprofv3 --kernel-trace --stats` for the profiler's per-kernel averages (the geometries are different instantiations, so one run separates them).
Needs the library of commit 1af148a (GP_TUNE_WG_WAVES = 25 and the W = 8 / 16 instantiations were removed again: not adopted).
Usage: python scripts/r06/wg_geometry.py [--points N] [--steps K] [--reps R]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

lib = gpa.load()


def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


N, STEPS, REPS = opt("--points", 1_000_000), opt("--steps", 400), opt("--reps", 3)
d = synthetic.make_c2_workload(N, 2_000_000, seed=42)
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
de = delta @ synthetic.expmap([0.002, -0.001, 0.003, 0.01, 0.02, -0.01])
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
pose_e = np.ascontiguousarray(de.T).reshape(1, 16).copy()
out = np.zeros((1, 122))
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
ref, ref_err = None, None
for rep in range(REPS):
    for waves in (4, 16, 8):
        arr = (C.c_void_p * 1)(f._h.value)
        batch, s = C.c_void_p(), C.c_void_p()
        lib.gp_stream_create(C.byref(s))
        _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 25, waves), "wg waves")
        lin = lib.gp_vgicp_batch_linearize
        pp, op = C.c_void_p(pose.ctypes.data), C.c_void_p(out.ctypes.data)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.2:
            _capi.check(lin(batch, pp, op), "linearize")
        eff = C.c_int(-1)
        lib.gp_vgicp_batch_get_tuning(batch, 26, C.byref(eff))
        if ref is None:
            ref = out.copy()
        rel = float(np.abs(out - ref).max() / np.abs(ref).max())
        mine = out.copy()
        lib.gp_vgicp_batch_device_times(batch, 1, None, None, None)
        t0 = time.perf_counter()
        for _ in range(STEPS):
            lin(batch, pp, op)
        wall = (time.perf_counter() - t0) / STEPS
        n, su, ku = C.c_double(), C.c_double(), C.c_double()
        lib.gp_vgicp_batch_device_times(batch, 0, C.byref(n), C.byref(su), C.byref(ku))
        alg = lib.gp_vgicp_batch_algorithmic_bytes(batch)
        again = bool(np.array_equal(out, mine))
        # error evaluation (fused), then the two-kernel forms of both
        e_f = np.zeros(1)
        _capi.check(lib.gp_vgicp_batch_compute_error(batch, pose.ctypes.data, pose_e.ctypes.data, e_f.ctypes.data), "error")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_FUSED_FINALIZE, 0), "two-kernel")
        two = np.zeros((1, 122))
        _capi.check(lin(batch, pp, C.c_void_p(two.ctypes.data)), "linearize two-kernel")
        e_t = np.zeros(1)
        _capi.check(lib.gp_vgicp_batch_compute_error(batch, pose.ctypes.data, pose_e.ctypes.data, e_t.ctypes.data), "error two-kernel")
        a, b_, c = C.c_float(), C.c_float(), C.c_float()
        _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 50, C.byref(a), C.byref(b_), C.byref(c)), "time")
        if ref_err is None:
            ref_err = float(e_f[0])
        print(json.dumps(dict(wg_waves=waves, effective_wg_waves=eff.value, rep=rep, points=N, step_us=round(wall * 1e6, 2), stream_us=round(su.value, 3), fused_us=round(ku.value, 3),
                              frac_whole_kernel=round(alg / (ku.value * 1e-6) / 8e12, 4) if ku.value else None, frac_streaming=round(alg / (su.value * 1e-6) / 8e12, 4) if su.value else None,
                              back_to_back_us=round(b_.value * 1e3, 3), reproducible=again, fused_equals_two_kernel=bool(np.array_equal(two, mine)), error_fused_equals_two_kernel=bool(e_f[0] == e_t[0]),
                              max_rel_diff_to_4_waves=rel, error_rel_diff_to_4_waves=abs(float(e_f[0]) - ref_err) / abs(ref_err), inliers=int(mine[0, 0]))), flush=True)
        lib.gp_vgicp_batch_destroy(batch)
        lib.gp_stream_destroy(s)
