"""Does the tile kernel run slower when it is launched after an idle gap?  Launches the C2 pass (tile + finalize kernel, asynchronous
issue + stream sync) with a host-side busy wait of G microseconds between passes; run under rocprofv3 --kernel-trace and feed the CSV to
the analysis at the bottom of scripts/r02/r02_idle_gap.sh."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
arr = (C.c_void_p * 1)(f._h.value); batch, s = C.c_void_p(), C.c_void_p()
lib.gp_stream_create(C.byref(s)); _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy(); out = np.zeros((1, 122))
for _ in range(20): lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
KEEP = float(os.environ.get("GP_PROBE_SPIN_US", "0"))  # > 0: a one-wave spinner of that many microseconds on a second stream before every pass
tune = _capi.load_tune() if KEEP > 0 else None
s2 = C.c_void_p(); lib.gp_stream_create(C.byref(s2))
for gap in [0, 2, 5, 10, 20, 50, 200, 1000]:
    for _ in range(40):
        if tune is not None:
            tune.gp_debug_spin(KEEP, s2)
        lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < gap: pass
    time.sleep(0.02)  # a long pause marks the boundary between gap settings in the trace
