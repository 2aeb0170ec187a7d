"""A/B of the workgroup -> tile map of the pipeline kernel (gp_debug_set_xcd_chunk) on the C2 workload: tile-kernel time (HIP events,
best of 3 x 50) for runs of c tiles dealt round robin to the XCDs; 0 = every XCD walks a contiguous eighth of the tile list."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

lib = gpa.load()
n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = synthetic.make_c2_workload(n_src, 2_000_000, seed=42)
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
ref = None
for chunk in [0, 1, 2, 4, 8, 16, 32, 64, 0]:
    _capi.check(lib.gp_debug_set_xcd_chunk(chunk), "chunk")
    arr = (C.c_void_p * 1)(f._h.value)
    batch, s = C.c_void_p(), C.c_void_p()
    lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
    out = np.zeros((1, 122))
    _capi.check(lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data), "lin")
    if ref is None:
        ref = out.copy()
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    best = (1e9, 0.0)
    for _ in range(3):
        _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 50, C.byref(a), C.byref(b), C.byref(c)), "time")
        best = min(best, (b.value, a.value))
    alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
    print(json.dumps(dict(source_points=n_src, xcd_chunk=chunk, tile_us=round(best[0] * 1e3, 2), pass_us=round(best[1] * 1e3, 2), frac=round(alg / (best[0] * 1e-3) / 8e12, 4),
                          max_abs_rel_vs_chunk0=float(np.abs(out - ref).max() / np.abs(ref).max()))), flush=True)
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)
lib.gp_debug_set_xcd_chunk(0)
