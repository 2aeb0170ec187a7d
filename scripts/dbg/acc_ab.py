"""A/B of the accumulate forms: python scripts/dbg/acc_ab.py [libname]; prints parity vs the oracle and the flat-vs-skewed-plan difference"""
import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gtsam_points_amd import _capi
if len(sys.argv) > 1:
    _capi.LIB_PATH = os.path.join(ROOT, "gtsam_points_amd", sys.argv[1])
import gtsam_points_amd as gpa
import oracle
from gtsam_points_amd import synthetic
lib = gpa.load()
BLOCKS = ["H_target", "H_source", "H_target_source", "b_target", "b_source"]
def lin(f, delta):
    rec = _capi.Linearized6()
    _capi.check(lib.gp_vgicp_factor_linearize(f._h, gpa.types._pose16(delta), C.byref(rec)), "lin")
    return gpa.LinearizedSystem6(rec)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for n, seed in [(1310757, 11), (1000000, 42), (300000, 3)]:
    d = synthetic.make_c2_workload(n, 500000, seed=seed)
    delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
    L = lin(gpa.IntegratedVGICPFactorGPU(0, 1, vm, src), delta)
    Lf = lin(gpa.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(5, 0), delta)
    L8 = lin(gpa.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(0, 8), delta)
    L3 = lin(gpa.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(0, 3), delta)  # f64 throughout
    om = oracle.OracleVoxelMap(0.5); om.insert(d["target_points"], d["target_covs"])
    Lo = oracle.OracleVGICPFactor(om, d["source_points"], d["source_covs"], oracle.max_threads()).linearize(delta)
    print(n, "vs oracle", {k: f"{rel(getattr(L, k), getattr(Lo, k)):.1e}" for k in BLOCKS}, flush=True)
    print(n, "flat vs oracle", {k: f"{rel(getattr(Lf, k), getattr(Lo, k)):.1e}" for k in BLOCKS}, flush=True)
    print(n, "r2 vs oracle", {k: f"{rel(getattr(L8, k), getattr(Lo, k)):.1e}" for k in BLOCKS}, flush=True)
    print(n, "f64 vs oracle", {k: f"{rel(getattr(L3, k), getattr(Lo, k)):.1e}" for k in BLOCKS}, flush=True)
    print(n, "skew vs flat", {k: f"{rel(getattr(L, k), getattr(Lf, k)):.1e}" for k in BLOCKS}, flush=True)
