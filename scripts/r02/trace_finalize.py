"""Timeline of the finalize kernel behind the C2 tile kernel (shader-clock stamps, gp_debug_set_trace_buffer): when the finalize
workgroup starts relative to the last tile workgroup of its XCD, and where its own time goes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

lib = gpa.load()
d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
arr = (C.c_void_p * 1)(f._h.value)
batch, s = C.c_void_p(), C.c_void_p()
lib.gp_stream_create(C.byref(s))
_capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
out = np.zeros((1, 122))
for _ in range(20):
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
trace = torch.zeros((2048, 16), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
_capi.check(lib.gp_debug_set_trace_buffer(C.c_void_p(trace.data_ptr())), "trace")
names = ["start", "partials summed (loads done)", "wave sums", "6x6 expansion", "record stored", "system fence + barrier", "flag stored"]
for rep in range(4):
    trace.zero_()
    torch.cuda.synchronize()
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
    torch.cuda.synchronize()
    raw = trace.cpu().numpy()
    fin = raw[2047]
    tiles = raw[:2047]
    tiles = tiles[tiles[:, 0] > 0]
    xcc = int(fin[9]) & 0xF
    same = tiles[(tiles[:, 9] & 0xF) == xcc]
    t0 = same[:, 0].min()
    print(f"rep {rep}: finalize on XCC {xcc}; tiles of that XCC: first start 0.00, last end {(same[:, 7].max() - t0) / 2100:.2f} us; finalize start {(fin[0] - t0) / 2100:.2f} us")
    print("   ", ", ".join(f"{n} +{(fin[k + 1] - fin[k]) / 2100:.2f}" for k, n in enumerate(names[1:])), f"| total {(fin[6] - fin[0]) / 2100:.2f} us")
lib.gp_debug_set_trace_buffer(None)
