"""Round 4: per-XCD / per-dispatch-round start and end times of the traced C2 tile kernel INSIDE a synchronous step (behind an idle queue), several steps.
Rows of the trace = tile index = xcd * (wgs / 8) + q; dispatch round = q // 32.  Times in us relative to the launch's first workgroup start (100 MHz clock)."""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
d = synthetic.make_c2_workload()
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
arr = (C.c_void_p * 1)(f._h.value); batch, s = C.c_void_p(), C.c_void_p()
lib.gp_stream_create(C.byref(s)); _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
for a in sys.argv[1:]:
    k, v = a.split("="); _capi.check(lib.gp_vgicp_batch_set_tuning(batch, int(k), int(v)), "tuning")
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy(); out = np.zeros((1, 122))
for _ in range(30): lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
acc = []
for rep in range(12):
    trace = torch.zeros((2048, 16), dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
    _capi.check(lib.gp_vgicp_batch_set_trace_buffer(batch, C.c_void_p(trace.data_ptr())), "trace")
    for _ in range(3): lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
    torch.cuda.synchronize()
    raw = trace.cpu().numpy()[:1024]
    lib.gp_vgicp_batch_set_trace_buffer(batch, None)
    rs, re_ = raw[:, 10].astype(np.float64) / 100, raw[:, 11].astype(np.float64) / 100
    t = raw[:, :8].astype(np.float64)
    s0 = rs.min()
    acc.append(np.stack([rs - s0, re_ - s0, (t[:, 1] - t[:, 0]), (t[:, 6] - t[:, 0]), (t[:, 7] - t[:, 6])], 1))
A = np.median(np.stack(acc), 0)  # [1024][5]
gx = 128
print(json.dumps(dict(start_max=round(float(A[:, 0].max()), 2), end_max=round(float(A[:, 1].max()), 2), end_p50=round(float(np.median(A[:, 1])), 2))))
for x in range(8):
    row = {}
    for r in range(4):
        sl = A[x * gx + 32 * r: x * gx + 32 * (r + 1)]
        row[f"round{r}"] = dict(start=round(float(np.median(sl[:, 0])), 2), end=round(float(np.median(sl[:, 1])), 2), end_max=round(float(sl[:, 1].max()), 2))
    print(json.dumps(dict(xcd=x, **row)))
