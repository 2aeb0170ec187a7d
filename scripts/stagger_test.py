import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
for variant in [2, 1]:
    lib.gp_debug_set_variant(variant)
    arr = (C.c_void_p * 1)(f._h.value); batch, s = C.c_void_p(), C.c_void_p(); lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
    for st in [0, 20, 40, 60, 80, 120, 160]:
        lib.gp_debug_set_stagger(st)
        a, b, c = C.c_float(), C.c_float(), C.c_float(); best = 1e9
        for _ in range(3):
            lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 50, C.byref(a), C.byref(b), C.byref(c)); best = min(best, b.value)
        print(f"variant {variant} stagger {st:4d}: tile {best*1e3:.2f} us", flush=True)
    lib.gp_vgicp_batch_destroy(batch)
