"""Round 6: gp_estimate_covariances (k = 10) on BASELINE configs[4]'s cloud (the 1 M-point C2 source), a few calls, with the work counters of the search
(gp_estimate_covariances_ex, counters_dev: queries, f32 / f64 distance evaluations, block entries, cells, far queries) read back -- the run the rocprofv3 passes of
scripts/r06/c5_pmc.sh profile.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import synthetic  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 7
SRC_ONLY = "src-only" in sys.argv  # the profiled passes: every covariance launch is then one of the config's cloud
d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
src, tgt = gpa.PointCloudGPU(d["source_points"]), gpa.PointCloudGPU(d["target_points"])
for _ in range(3):
    gpa.estimate_covariances_gpu(src, 10)
    if not SRC_ONLY:
        gpa.estimate_covariances_gpu(tgt, 10)
ts = []
for _ in range(REPS):
    if not SRC_ONLY:
        gpa.estimate_covariances_gpu(tgt, 10)  # (alternating, as bench.py does)
    torch.cuda.synchronize()
    t = time.perf_counter()
    gpa.estimate_covariances_gpu(src, 10)
    ts.append(time.perf_counter() - t)
counters = torch.zeros(8 + 8 * (2 * ((1_000_000 + 63) // 64) + 8), dtype=torch.int64, device="cuda")
gpa.estimate_covariances_gpu(src, 10, counters=counters)
torch.cuda.synchronize()
c = counters[:8].cpu().numpy()
print(json.dumps(dict(points=1_000_000, k=10, ms_median=round(float(np.median(ts)) * 1e3, 4), ms_min=round(float(np.min(ts)) * 1e3, 4),
                      counters=dict(queries=int(c[0]), f32_distance_evaluations=int(c[1]), f64_distance_evaluations=int(c[2]), block_entries=int(c[3]), cells=int(c[4]), far_queries=int(c[5])))))
