"""Timeline of the rolling-DMA tile kernel in the BATCHED regime (F factors over the same 1 M-point pair -> F x 977 tiles,
several residency rounds): per-workgroup s_memtime stamps -> phase durations in ticks.  python scripts/trace_batch.py 1 4"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = gpa.load()
d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
fs = [gpa.IntegratedVGICPFactorGPU(0, 1, vm, src) for _ in range(F)]
_capi.check(lib.gp_debug_set_variant(variant), "variant")
arr = (C.c_void_p * F)(*[f._h.value for f in fs]); batch, s = C.c_void_p(), C.c_void_p(); lib.gp_stream_create(C.byref(s))
_capi.check(lib.gp_vgicp_batch_create(arr, F, s, C.byref(batch)), "batch")
pose = np.tile(np.ascontiguousarray(delta.T).reshape(1, 16), (F, 1)).copy(); out = np.zeros((F, 122))
for _ in range(3): lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
a, b, c = C.c_float(), C.c_float(), C.c_float()
lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 20, C.byref(a), C.byref(b), C.byref(c))
print(f"variant {variant} F {F}: tile kernel {b.value*1e3:.1f} us = {b.value*1e3/F:.2f} us per 1M points")
T = 1024 * F
trace = torch.zeros((T, 8), dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
_capi.check(lib.gp_debug_set_trace_buffer(C.c_void_p(trace.data_ptr())), "trace")
lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data); torch.cuda.synchronize()
lib.gp_debug_set_trace_buffer(None)
t = trace.cpu().numpy().astype(np.float64); t = t[t[:, 0] > 0]
names = ["start", "chunk0_landed", "gather0_landed", "step0_done", "gather1_landed", "step1_done", "steps_done", "end"]
dur = np.diff(t, axis=1)
print("workgroups traced:", len(t), " block lifetime ticks median", np.median(t[:, 7] - t[:, 0]))
for k in range(7):
    print(f"phase {names[k]:>14s} -> {names[k+1]:14s}: median {np.median(dur[:,k]):8.0f}  p10 {np.percentile(dur[:,k],10):8.0f}  p90 {np.percentile(dur[:,k],90):8.0f} ticks")
