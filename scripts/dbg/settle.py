"""how long until the synchronous step settles: continuous stepping from a cold start, per 100 ms window the mean / p90 / max host step"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
d = synthetic.make_c2_workload(1_000_000, 2_000_000, seed=42)
stream = torch.cuda.Stream()
sptr = C.c_void_p(stream.cuda_stream)
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy(); out = np.zeros((1, 122))
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src, stream=sptr)
arr = (C.c_void_p * 1)(f._h.value); batch = C.c_void_p()
_capi.check(lib.gp_vgicp_batch_create(arr, 1, sptr, C.byref(batch)), "batch")
lin = lib.gp_vgicp_batch_linearize; pp, op = C.c_void_p(pose.ctypes.data), C.c_void_p(out.ctypes.data)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
time.sleep(2.0)
T0 = time.perf_counter()
for w in range(30):
    ts = []
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.1:
        t = time.perf_counter(); lin(batch, pp, op); ts.append(time.perf_counter() - t)
        if mode == "sync" and len(ts) % 25 == 0:
            torch.cuda.synchronize()
    a = np.array(ts) * 1e6
    print(f"{mode} window {w:2d} (+{time.perf_counter()-T0:4.1f} s): steps {len(a):5d} mean {a.mean():6.2f} p50 {np.median(a):6.2f} p90 {np.percentile(a,90):6.2f} max {a.max():7.1f} us", flush=True)
