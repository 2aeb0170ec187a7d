import ctypes as C, time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
d = synthetic.make_c2_workload()
pts = torch.from_numpy(d["source_points"]).cuda()
n = pts.shape[0]
torch.cuda.synchronize()
for _ in range(3):
    t = time.perf_counter(); g = C.c_void_p(); _capi.check(lib.gp_point_grid_create(C.c_void_p(pts.data_ptr()), n, 0.125, None, C.byref(g)), "grid"); t1 = time.perf_counter()
    lib.gp_point_grid_destroy(g); t2 = time.perf_counter()
    covs = torch.empty((n, 9), dtype=torch.float32, device="cuda"); short = C.c_int()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    _capi.check(lib.gp_estimate_covariances(C.c_void_p(pts.data_ptr()), n, 10, 0.0, C.c_void_p(covs.data_ptr()), C.byref(short), None), "cov"); t4 = time.perf_counter()
    print(f"grid create {1e3*(t1-t):.2f} ms, destroy {1e3*(t2-t1):.2f} ms, estimate_covariances total {1e3*(t4-t3):.2f} ms")
