"""Round 6: GaussianVoxelMapGPU::insert of the 2 M-point C2 target at 0.5 m, N builds -- the run scripts/r06/map_build_pmc.sh profiles (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
in separate passes; --kernel-trace --stats for the durations).  Prints the wall per build."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
d = synthetic.make_c2_workload(1000, 2_000_000, seed=42)
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
ts = []
for _ in range(N):
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    vm.insert(tgt)
    ts.append(time.perf_counter() - t)
print(json.dumps(dict(points=tgt.size(), builds=N, ms_median=round(float(np.median(ts[2:])) * 1e3, 4), ms_min=round(float(np.min(ts[2:])) * 1e3, 4), num_voxels=int(vm.voxelmap_info.num_voxels))))
