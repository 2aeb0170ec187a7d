"""Round 4: voxel-map build (gp_voxelmap_insert) of the 2 M-point C2 target at 0.5 m and k-NN structure + covariances of the 1 M-point source: wall per call, median of 20."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gtsam_points_amd as gpa
from gtsam_points_amd import synthetic
d = synthetic.make_c2_workload(1_000_000, 2_000_000, seed=42)
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"])
ts = []
for i in range(25):
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    torch.cuda.synchronize()
    t = time.perf_counter(); vm.insert(tgt); ts.append(time.perf_counter() - t)
info = vm.voxelmap_info
tc = []
for i in range(12):
    torch.cuda.synchronize(); t = time.perf_counter(); gpa.estimate_covariances_gpu(src, 10); tc.append(time.perf_counter() - t)
print(json.dumps(dict(map_build_ms_median=round(float(np.median(ts[5:])) * 1e3, 4), map_build_ms_min=round(float(np.min(ts[5:])) * 1e3, 4), first_build_ms=round(ts[0] * 1e3, 3),
                      num_voxels=info.num_voxels, bytes_in=96_000_000, frac_of_8TBs=round(96e6 / float(np.median(ts[5:])) / 8e12, 4),
                      covariances_ms_median=round(float(np.median(tc[2:])) * 1e3, 4))))
