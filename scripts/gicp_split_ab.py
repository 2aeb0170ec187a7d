import ctypes as C, os, sys, time, json
import numpy as np, torch
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.getcwd())
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic, types
lib = gpa.load()
d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
fg = gpa.IntegratedGICPFactorGPU(0, 1, tgt, src)
base = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
poses = [base @ synthetic.expmap(np.random.default_rng(i).uniform(-1e-4, 1e-4, 6)) for i in range(12)]
for p in poses[:2]: fg.linearize_delta(p)
ts = []
for p in poses[2:]:
    t = time.perf_counter(); L = fg.linearize_delta(p); ts.append(time.perf_counter() - t)
out = C.c_double(); te = []
for p in poses[2:]:
    t = time.perf_counter(); lib.gp_gicp_factor_compute_error(fg._h, types._pose16(poses[-1]), types._pose16(p), C.byref(out)); te.append(time.perf_counter() - t)
print(json.dumps(dict(split=os.environ.get("GP_GICP_SPLIT", "1"), gicp_linearize_ms_distinct_poses=round(float(np.median(ts)) * 1e3, 4), gicp_error_ms=round(float(np.median(te)) * 1e3, 4), inliers=int(L.num_inliers))))
