"""what the driver's K = 20 timed region is made of: per-step wall of the 20 steps behind the opening synchronisation, and the closing torch.cuda.synchronize()"""
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
T00 = time.perf_counter()
lin = lib.gp_vgicp_batch_linearize; pp, op = C.c_void_p(pose.ctypes.data), C.c_void_p(out.ctypes.data)
if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]))
for trial in range(6):
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        lin(batch, pp, op)
    for _ in range(5):
        lin(batch, pp, op)
    ta = time.perf_counter(); torch.cuda.synchronize(); tb = time.perf_counter()
    ts = []
    t0 = time.perf_counter()
    for _ in range(20):
        t = time.perf_counter(); lin(batch, pp, op); ts.append(time.perf_counter() - t)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"trial {trial} (+{time.perf_counter()-T00:.2f} s): opening sync {1e6*(tb-ta):.1f} us; steps us:", " ".join(f"{x*1e6:.1f}" for x in ts), f"; loop {1e6*(t1-t0):.1f} us; closing sync {1e6*(t2-t1):.1f} us; per step incl. closing sync {1e6*(t2-t0)/20:.2f}", flush=True)
