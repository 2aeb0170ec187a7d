import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gtsam_points_amd as gpa
import oracle
from gtsam_points_amd import synthetic, _capi
lib = gpa.load()
def lin(f, delta):
    rec = _capi.Linearized6()
    _capi.check(lib.gp_vgicp_factor_linearize(f._h, gpa.types._pose16(delta), C.byref(rec)), "lin")
    return gpa.LinearizedSystem6(rec)
for n in [64, 256, 1000, 20000, 150000]:
    d = synthetic.make_pair(n, 200000, seed=5)
    d["source_normals"] = d["source_normals"].copy(); d["source_normals"][::3] *= -1.0
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
    src = gpa.PointCloudGPU(d["source_points"], d["source_covs"], normals=d["source_normals"])
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
    delta = d["T_true"]
    res = {}
    for mirror in (1, 0):
        for pol in (1, 2):
            f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(21, mirror).set_tuning(1, pol)
            f.set_enable_surface_validation(True)
            res[(mirror, pol)] = lin(f, delta)
    f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(0, 8); f.set_enable_surface_validation(True)
    res["r2"] = lin(f, delta)
    base = res[(0, 1)]
    print(n, {k: (v.num_inliers, float(np.abs(v.H_source - base.H_source).max() / np.abs(base.H_source).max())) for k, v in res.items()}, flush=True)
