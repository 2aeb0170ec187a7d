"""Covariance-estimation time vs the finest grid cell size (3 levels, x4 each) on the 1 M-point bench source cloud."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
d = synthetic.make_c2_workload()
pts = torch.from_numpy(d["source_points"]).cuda(); n = pts.shape[0]
covs = torch.empty((n, 9), dtype=torch.float32, device="cuda"); short = C.c_int()
ref = None
for cell in [float(c) for c in sys.argv[1:]] or [0.25, 0.125]:
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        _capi.check(lib.gp_estimate_covariances(C.c_void_p(pts.data_ptr()), n, 10, cell, C.c_void_p(covs.data_ptr()), C.byref(short), None), "cov")
        best = min(best, time.perf_counter() - t)
    c = covs.cpu().numpy()
    if ref is None: ref = c
    print(f"cell {cell:7.4f}: {best*1e3:6.2f} ms, short {short.value}, max |dcov| vs first {np.abs(c - ref).max():.2e}", flush=True)
