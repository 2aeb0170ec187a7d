"""Soak test: many linearise / error / covariance / overlap / merge / solve calls; device memory must not creep."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
NS, NT = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (120000, 200000)  # >= 393216 source points: the second-generation tile kernel
d = synthetic.make_pair(NS, NT, seed=5)
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5); vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src); f.set_enable_offloading(True)
fset = gpa.NonlinearFactorSetGPU(); fset.add(f)
values = {0: np.eye(4), 1: d["T_true"]}
def free_mb():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
fset.linearize(values); f.linearize(values)
gpa.estimate_covariances_gpu(src, 10); gpa.merge_frames_gpu([np.eye(4), d["T_true"]], [tgt, src], 0.3)
m0 = free_mb(); t0 = time.time()
for it in range(3000):
    fset.linearize(values); f.linearize(values); fset.error(values); f.error(values)
    if it % 100 == 0:
        gpa.estimate_covariances_gpu(src, 10)
        gpa.overlap_gpu([vm, vm], src, [np.eye(4), d["T_true"]])
        gpa.merge_frames_gpu([np.eye(4), d["T_true"]], [tgt, src], 0.3)
        rec = gpa.linearize_on_device([f], values); gpa.DenseLinearSystemGPU(1, [(-1, 0)]).build(rec, lam=1e-3).solve()
        gpa.SparseLinearSystemGPU(1, [(-1, 0)]).build(rec, lam=1e-3).solve()
        fg = gpa.IntegratedGICPFactorGPU(0, 1, tgt, src); fg.linearize_delta(d["T_true"])
    if it % 500 == 0:
        src.offload_gpu(); vm.offload_gpu()
m1 = free_mb()
print(f"3000 iterations in {time.time()-t0:.1f} s; free device memory {m0:.0f} -> {m1:.0f} MiB (delta {m0-m1:+.0f})")
assert m0 - m1 < 64, "device memory creeps"
print("SOAK_OK")
