"""Runs only the linearise pass of the C2 workload for one tile-kernel variant (for rocprofv3 --pmc passes)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = gpa.load()
d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
_capi.check(lib.gp_debug_set_variant(variant), "variant")
arr = (C.c_void_p * 1)(f._h.value)
batch, s = C.c_void_p(), C.c_void_p()
lib.gp_stream_create(C.byref(s))
_capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
a, b, c = C.c_float(), C.c_float(), C.c_float()
_capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, iters, C.byref(a), C.byref(b), C.byref(c)), "time")
print(f"variant {variant}: tile {b.value*1e3:.2f} us, finalize {c.value*1e3:.2f} us, pass {a.value*1e3:.2f} us")
