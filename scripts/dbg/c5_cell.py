"""covariances per call against the search cell size (exactness does not depend on it)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gtsam_points_amd as gpa
from gtsam_points_amd import synthetic
d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
ref = None
for name in ("source_points", "target_points"):
    for cell in (0.25, 0.2, 0.22, 0.28, 0.3, 0.32, 0.36, 0.4, 0.25):
        src = gpa.PointCloudGPU(d[name])
        for _ in range(3):
            gpa.estimate_covariances_gpu(src, 10, cell_size=cell)
        ts = []
        for _ in range(9):
            torch.cuda.synchronize(); t = time.perf_counter(); gpa.estimate_covariances_gpu(src, 10, cell_size=cell); ts.append(time.perf_counter() - t)
        cov = src.download("covs")
        if cell == 0.25 and ref is None:
            ref = {}
        if name not in ref:
            ref[name] = cov
        print(json.dumps(dict(cloud=name, cell=cell, ms_median=round(float(np.median(ts)) * 1e3, 4), max_abs_diff_vs_025=float(np.abs(cov - ref[name]).max()))), flush=True)
