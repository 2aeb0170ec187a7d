"""Round-5 A/B on the GPU box (VERDICT r04 #1b / #1c): the C2 headline step as bench.py runs it (synchronous fused call) with the measurement instantiations of the stream
kernel (GP_TUNE_EXPERIMENT: 1 = block-grid warm-up behind the first request, 2 = R C_A R^T in f32 -- breaks parity, timing only) against the product kernel, alternating,
by the kernel's own 100 MHz stamps inside the steps (streaming part, whole fused kernel), host wall per step and the tile kernel back to back (HIP events).
One JSON object per line.  Usage: python scripts/r05_sweep.py [--points N] [--steps K] [--reps R]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
out = np.zeros((1, 122))
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
ref = None
NAMES = {0: "product kernel", 1: "block-grid warm-up", 2: "f32 covariance rotation (parity broken: timing only)"}
for rep in range(REPS):
    for exp in (0, 1, 2):
        arr = (C.c_void_p * 1)(f._h.value)
        batch, s = C.c_void_p(), C.c_void_p()
        lib.gp_stream_create(C.byref(s))
        _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_EXPERIMENT, exp), "experiment")
        lin = lib.gp_vgicp_batch_linearize
        pp, op = C.c_void_p(pose.ctypes.data), C.c_void_p(out.ctypes.data)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.2:
            lin(batch, pp, op)
        if ref is None:
            ref = out.copy()
        rel = float(np.abs(out - ref).max() / np.abs(ref).max())
        lib.gp_vgicp_batch_device_times(batch, 1, None, None, None)
        t0 = time.perf_counter()
        for _ in range(STEPS):
            lin(batch, pp, op)
        wall = (time.perf_counter() - t0) / STEPS
        n, su, ku = C.c_double(), C.c_double(), C.c_double()
        lib.gp_vgicp_batch_device_times(batch, 0, C.byref(n), C.byref(su), C.byref(ku))
        alg = lib.gp_vgicp_batch_algorithmic_bytes(batch)
        print(json.dumps(dict(experiment=exp, name=NAMES[exp], rep=rep, points=N, step_us=round(wall * 1e6, 2), stream_us=round(su.value, 3), fused_us=round(ku.value, 3),
                              frac_whole_kernel=round(alg / (ku.value * 1e-6) / 8e12, 4) if ku.value else None, frac_streaming=round(alg / (su.value * 1e-6) / 8e12, 4) if su.value else None,
                              bit_equal_to_product=bool(np.array_equal(ref, out)), max_rel_diff_to_product=rel)), flush=True)
        lib.gp_vgicp_batch_destroy(batch)
        lib.gp_stream_destroy(s)
