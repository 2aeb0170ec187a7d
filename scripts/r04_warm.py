"""Round 4: does the C2 step depend on how long the device has been busy?  The bench's timed region follows 20 warm-up steps (0.4 ms) behind seconds of host-side set-up.
In-step streaming time (the kernel's own stamps) of consecutive blocks of 200 steps from a cold start, then after 0.3 s of back-to-back launches, then after 2 s idle."""
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
arr = (C.c_void_p * 1)(f._h.value); b, s = C.c_void_p(), C.c_void_p()
lib.gp_stream_create(C.byref(s)); _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(b)), "batch")
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy(); out = np.zeros((1, 122))
pp, op = C.c_void_p(pose.ctypes.data), C.c_void_p(out.ctypes.data)
lin = lib.gp_vgicp_batch_linearize
def block(label, steps=200):
    lib.gp_vgicp_batch_device_times(b, 1, None, None, None)
    t0 = time.perf_counter()
    for _ in range(steps): lin(b, pp, op)
    wall = (time.perf_counter() - t0) / steps
    n, su, ku = C.c_double(), C.c_double(), C.c_double()
    lib.gp_vgicp_batch_device_times(b, 0, C.byref(n), C.byref(su), C.byref(ku))
    print(json.dumps(dict(block=label, step_us=round(wall * 1e6, 2), stream_us=round(su.value, 3), frac=round(56028980 / (su.value * 1e-6) / 8e12, 4))), flush=True)
time.sleep(2.0)
for i in range(20): lin(b, pp, op)
for k in range(6): block(f"cold+{k}")
a_, b_, c_ = C.c_float(), C.c_float(), C.c_float()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    lib.gp_vgicp_batch_time_linearize(b, pp, 200, C.byref(a_), C.byref(b_), C.byref(c_))
for k in range(4): block(f"after 0.3 s busy +{k}")
time.sleep(2.0)
for k in range(4): block(f"after 2 s idle +{k}")
for k in range(3): block(f"2000-step block {k}", 2000)
