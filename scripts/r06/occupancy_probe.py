"""Round 6, TIMING ONLY: would a fifth wave per SIMD help the stream kernel where more workgroups than fit are queued (batches, the 8 M-point source)?  --lib names a build of
the library (the probe build libgtsam_points_hip_occ5.so gives the packed instantiations 8 KB of LDS per wave by shrinking the reduction's transposition buffer BELOW what it
needs -- its sums are wrong, its instruction stream and memory traffic are the product's -- and asks for five waves per SIMD).  Prints kernel times of C3's batched call and of
the 8 M-point source."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from gtsam_points_amd import _capi  # noqa: E402

lib_name = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else "libgtsam_points_hip.so"
_capi.LIB_PATH = os.path.join(ROOT, "gtsam_points_amd", lib_name)
import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import synthetic  # noqa: E402

lib = gpa.load()


def kernel_ms(factors, poses, iters=50, max_wgs=None):
    arr = (C.c_void_p * len(factors))(*[f._h.value for f in factors])
    batch, s = C.c_void_p(), C.c_void_p()
    lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, len(factors), s, C.byref(batch)), "batch")
    if max_wgs:
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_MAX_WORKGROUPS, max_wgs), "max workgroups")
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    best = 1e9
    for _ in range(3):
        _capi.check(lib.gp_vgicp_batch_time_linearize(batch, poses.ctypes.data, iters, C.byref(a), C.byref(b), C.byref(c)), "time")
        best = min(best, b.value)
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)
    return round(best * 1e3, 2)


g = synthetic.make_c3_graph()
clouds = [gpa.PointCloudGPU(p, c) for p, c in g["clouds"]]
maps = []
for c in clouds:
    m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
    m.insert(c)
    maps.append(m)
factors = [gpa.IntegratedVGICPFactorGPU(t, s_, maps[t], clouds[s_]) for t, s_ in g["pairs"]]
poses = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in g["deltas"]]).copy()
row = dict(lib=lib_name, c3_kernel_us=kernel_ms(factors, poses))
d = synthetic.make_c2_workload(8_000_000, 2_000_000, seed=42)
tgt, src = gpa.PointCloudGPU(d["target_points"], d["target_covs"]), gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
pose = np.ascontiguousarray((d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])).T).reshape(1, 16).copy()
row["big_source_us_1024"] = kernel_ms([f], pose, 20, 1024)
if "occ5" in lib_name:
    row["big_source_us_1280"] = kernel_ms([f], pose, 20, 1280)
print(json.dumps(row), flush=True)
