"""Round-4 sweep on the GPU box: the C2 headline step as bench.py runs it (synchronous fused call), per tuning case, measured by the kernel's own
100 MHz stamps inside the steps (gp_vgicp_batch_device_times: streaming part and whole fused kernel) + host wall per step + the tile kernel back to back
(HIP events).  One JSON object per line.
Usage: python scripts/r04_sweep.py "mirror:balance:max_wgs[:policy],..." [--points N] [--steps K]"""
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

if "--lib" in sys.argv:  # (A/B of a differently built library: the file name under gtsam_points_amd/)
    _capi.LIB_PATH = os.path.join(ROOT, "gtsam_points_amd", sys.argv[sys.argv.index("--lib") + 1])
lib = gpa.load()
args = [a for a in sys.argv[1:] if not a.startswith("--") and not a.endswith(".so")]


def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


N, STEPS = opt("--points", 1_000_000), opt("--steps", 300)
cases = [tuple(int(x) for x in c.split(":")) for c in (args[0] if args else "1:-1:1024,0:-1:1024").split(",")]
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
for rep in range(2):
    for case in cases:
        mirror, balance, wgs = case[:3]
        arr = (C.c_void_p * 1)(f._h.value)
        batch, s = C.c_void_p(), C.c_void_p()
        lib.gp_stream_create(C.byref(s))
        _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_SOURCE_MIRROR, mirror), "mirror")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_BALANCE, balance), "balance")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_MAX_WORKGROUPS, wgs), "wgs")
        if len(case) > 3:
            _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_SOURCE_POLICY, case[3]), "policy")
        if len(case) > 4 and case[4] == 0:  # equal XCD shares (the library's table otherwise)
            for x in range(8):
                _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_XCD_WEIGHT_0 + x, 1000), "xcd weight")
        lin = lib.gp_vgicp_batch_linearize
        pp, op = C.c_void_p(pose.ctypes.data), C.c_void_p(out.ctypes.data)
        for _ in range(30):
            lin(batch, pp, op)
        if ref is None:
            ref = out.copy()
        same = bool(np.array_equal(ref, out))
        close = float(np.abs(out - ref).max() / np.abs(ref).max())
        lib.gp_vgicp_batch_device_times(batch, 1, None, None, None)
        t0 = time.perf_counter()
        for _ in range(STEPS):
            lin(batch, pp, op)
        wall = (time.perf_counter() - t0) / STEPS
        n, su, ku = C.c_double(), C.c_double(), C.c_double()
        lib.gp_vgicp_batch_device_times(batch, 0, C.byref(n), C.byref(su), C.byref(ku))
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        best = 1e9
        for _ in range(3):
            _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 50, C.byref(a), C.byref(b), C.byref(c)), "time")
            best = min(best, b.value)
        alg = lib.gp_vgicp_batch_algorithmic_bytes(batch)
        print(json.dumps(dict(case=":".join(map(str, case)), rep=rep, points=N, step_us=round(wall * 1e6, 2), stream_us=round(su.value, 3), fused_us=round(ku.value, 3),
                              frac_in_step=round(alg / (su.value * 1e-6) / 8e12, 4) if su.value else None, b2b_us=round(best * 1e3, 3), frac_b2b=round(alg / (best * 1e-3) / 8e12, 4),
                              bit_equal_to_first=same, max_rel_diff=close)), flush=True)
        lib.gp_vgicp_batch_destroy(batch)
        lib.gp_stream_destroy(s)
