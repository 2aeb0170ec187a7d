"""alternating clouds through gp_estimate_covariances: wall per call and the library's own split (GP_KNN_DEBUG)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gtsam_points_amd as gpa
from gtsam_points_amd import synthetic
d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
tgt, src = gpa.PointCloudGPU(d["target_points"]), gpa.PointCloudGPU(d["source_points"])
for fr in (tgt, src):
    gpa.estimate_covariances_gpu(fr, 10)
for name, fr in (("tgt", tgt), ("src", src), ("tgt", tgt), ("src", src), ("src", src), ("src", src), ("tgt", tgt), ("tgt", tgt)):
    torch.cuda.synchronize()
    t = time.perf_counter()
    gpa.estimate_covariances_gpu(fr, 10)
    print(name, round((time.perf_counter() - t) * 1e3, 4), "ms", flush=True)
