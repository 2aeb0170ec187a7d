"""C5 timing on the GPU box: gp_estimate_covariances (k = 10) of the 1 M-point C2 source, wall time per call (median of 9) and, under rocprofv3, the kernels'
own durations; GP_COV_FAST=0 selects the exact list throughout (round 2), GP_COV_FAST_WAVES=3 the 160-register build of the fast kernel.
Prints one JSON line; with `--ref <npy>` the covariances are compared with / written to a file so that two builds can be diffed."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import synthetic  # noqa: E402

d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
src = gpa.PointCloudGPU(d["source_points"])
for _ in range(3):
    gpa.estimate_covariances_gpu(src, 10)
ts = []
for _ in range(9):
    torch.cuda.synchronize()
    t = time.perf_counter()
    short = gpa.estimate_covariances_gpu(src, 10)
    ts.append(time.perf_counter() - t)
cov = src.download("covs").astype(np.float64)
out = dict(fast=os.environ.get("GP_COV_FAST", "1"), waves=os.environ.get("GP_COV_FAST_WAVES", "4"), ms_median=round(float(np.median(ts)) * 1e3, 4), ms_min=round(float(np.min(ts)) * 1e3, 4),
           short=int(short))
if "--ref" in sys.argv:
    path = sys.argv[sys.argv.index("--ref") + 1]
    if os.path.exists(path):
        ref = np.load(path)
        rel = np.linalg.norm((cov - ref).reshape(len(cov), -1), axis=1) / np.linalg.norm(ref.reshape(len(cov), -1), axis=1)
        out.update(vs_ref_max=float(rel.max()), vs_ref_frac_gt_1e6=float((rel > 1e-6).mean()))
    else:
        np.save(path, cov)
print(json.dumps(out), flush=True)
