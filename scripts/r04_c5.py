"""Round 4, C5: gp_estimate_covariances (k = 10) of the 1 M-point C2 source per GP_TUNE_KNN_STRUCTURE value: wall time per call (median of 9, GP_KNN_DEBUG split into
structure build / search when set), identical results across the structures (covariances bit for bit).  Also the kitti-like far-field heavy case.
Usage: python scripts/r04_c5.py [structures, e.g. 0,5]"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gtsam_points_amd as gpa
from gtsam_points_amd import synthetic
structures = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,5").split(",")]
d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
ref = None
for rep in range(2):
    for st in structures:
        src = gpa.PointCloudGPU(d["source_points"])
        for _ in range(3):
            gpa.estimate_covariances_gpu(src, 10, structure=st)
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            t = time.perf_counter()
            short = gpa.estimate_covariances_gpu(src, 10, structure=st)
            ts.append(time.perf_counter() - t)
        cov = src.download("covs")
        if ref is None:
            ref = cov
        print(json.dumps(dict(structure=st, rep=rep, ms_median=round(float(np.median(ts)) * 1e3, 4), ms_min=round(float(np.min(ts)) * 1e3, 4), short=int(short),
                              bit_equal_to_first=bool(np.array_equal(ref, cov)))), flush=True)
