"""Round 5: today's VGICP error evaluation on C2 -- host wall per gp_vgicp_batch_compute_error and per linearise, the same calls' kernels under
rocprofv3 --stats when run under it.  One JSON line."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
d = synthetic.make_c2_workload(1_000_000, 2_000_000, seed=42)
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
arr = (C.c_void_p * 1)(f._h.value); batch = C.c_void_p(); s = C.c_void_p(); lib.gp_stream_create(C.byref(s))
_capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015]); de = delta @ synthetic.expmap([1e-4, 2e-4, -1e-4, 0.005, 0.002, -0.003])
pl = np.ascontiguousarray(delta.T).reshape(1, 16).copy(); pe = np.ascontiguousarray(de.T).reshape(1, 16).copy()
rec = np.zeros((1, 122)); err = np.zeros(1)
lin, ev = lib.gp_vgicp_batch_linearize, lib.gp_vgicp_batch_compute_error
a, b, c, e = C.c_void_p(pl.ctypes.data), C.c_void_p(pe.ctypes.data), C.c_void_p(rec.ctypes.data), C.c_void_p(err.ctypes.data)
t = time.perf_counter()
while time.perf_counter() - t < 0.3:
    lin(batch, a, c); ev(batch, a, b, e)
res = {}
for name, call in (("linearize", lambda: lin(batch, a, c)), ("compute_error", lambda: ev(batch, a, b, e)), ("linearize + 2 x compute_error (LM cadence)", lambda: (lin(batch, a, c), ev(batch, a, b, e), ev(batch, a, b, e)))):
    for _ in range(50): call()
    t0 = time.perf_counter()
    for _ in range(500): call()
    res[name] = round((time.perf_counter() - t0) / 500 * 1e6, 2)
print(json.dumps(dict(workload="C2: 1 M source points vs the 2 M-point map @0.5 m", host_us_per_call=res, error=float(err[0]), linearise_error=float(rec[0, 1]))))
