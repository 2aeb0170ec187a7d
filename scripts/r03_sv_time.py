"""Surface validation at the headline size: C2 (1 M source points with normals vs the 2 M-point map) through the default kernel with and
without the normals row -- tile-kernel time (HIP events, back to back), algorithmic bytes (48 vs 60 B per source point) and the fraction of 8 TB/s; the
round-2 kernel (family 8), which validating factors fell back to, beside it.  One JSON object per line."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

lib = gpa.load()
d = synthetic.make_c2_workload()
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"], d["source_covs"], normals=d["source_normals"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
pose = np.ascontiguousarray((d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])).T).reshape(1, 16).copy()
out = np.zeros((1, 122))
for family in (12, 8):
    for sv in (0, 1):
        f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
        f.set_tuning(0, family)
        f.set_enable_surface_validation(bool(sv))
        arr = (C.c_void_p * 1)(f._h.value)
        batch, s = C.c_void_p(), C.c_void_p()
        lib.gp_stream_create(C.byref(s))
        _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 0, family), "kernel")
        best = 1e9
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        for _ in range(3):
            _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 50, C.byref(a), C.byref(b), C.byref(c)), "time")
            best = min(best, b.value)
        _capi.check(lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data), "linearize")
        alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
        eff = C.c_int()
        lib.gp_vgicp_batch_get_tuning(batch, 6, C.byref(eff))
        print(json.dumps(dict(family=family, effective_family=eff.value, surface_validation=sv, tile_us=round(best * 1e3, 2), algorithmic_bytes=alg,
                              frac=round(alg / (best * 1e-3) / 8e12, 4), inliers=float(out[0, 0]))), flush=True)
        lib.gp_vgicp_batch_destroy(batch)
        lib.gp_stream_destroy(s)
