"""Timeline of the pipeline tile kernel: per-workgroup s_memtime stamps -> phase durations (tuning tool).
Only the per-workgroup DIFFERENCES are meaningful (every XCD has its own counter)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 1
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
out = np.zeros((1, 122))
for _ in range(5):
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
T = 1024
trace = torch.zeros((T, 8), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
_capi.check(lib.gp_debug_set_trace_buffer(C.c_void_p(trace.data_ptr())), "trace")
lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
torch.cuda.synchronize()
lib.gp_debug_set_trace_buffer(None)
t = trace.cpu().numpy().astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
rel = (t - t0) / 2100.0  # s_memtime counts shader clocks (~2.1 GHz under this load); counters of different XCDs are not synchronised
names = ["start", "chunk0_landed", "keys0_landed", "step0_done", "keys1_landed", "step1_done", "steps_done", "end"]
if variant >= 3:
    names = ["start", "chunk0_landed", "iter0_top", "iter1_top", "iter2_top", "iter3_top", "loop_done", "end"]
print("workgroups traced:", len(t))
for k, n in enumerate(names):
    c = rel[:, k]
    print(f"{n:14s} min {c.min():7.2f}  p10 {np.percentile(c,10):7.2f}  median {np.median(c):7.2f}  p90 {np.percentile(c,90):7.2f}  max {c.max():7.2f} us")
dur = np.diff(rel[:, :len(names)], axis=1)
for k in range(len(names) - 1):
    print(f"phase {names[k]:>13s} -> {names[k+1]:13s}: median {np.median(dur[:,k]):6.2f}  p90 {np.percentile(dur[:,k],90):6.2f} us")
# dispatch skew per clock domain: s_memtime counters are not synchronised across the chip, so workgroups are clustered by
# their start stamp (gaps > 50 us separate domains) and skew / lifetime are taken inside each cluster
raw = trace.cpu().numpy().astype(np.float64)
raw = raw[raw[:, 0] > 0]
raw = raw[np.argsort(raw[:, 0])]
cuts = np.nonzero(np.diff(raw[:, 0]) > 50 * 2100)[0] + 1
for k, r in enumerate(np.split(raw, cuts)):
    s0 = r[:, 0].min()
    print(f"clock domain {k}: {len(r):4d} workgroups, start skew {(r[:,0].max()-s0)/2100:5.2f} us, last end {(r[:,7].max()-s0)/2100:5.2f} us, "
          f"median lifetime {np.median(r[:,7]-r[:,0])/2100:5.2f} us")
