"""A/B the tile-kernel variants on the GPU box: kernel time (HIP events) + parity vs the oracle.  Tuning tool."""
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
import oracle  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

lib = gpa.load()
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2".split(","))]
BLOCKS = ["H_target", "H_source", "H_target_source", "b_target", "b_source"]


def run_case(name, d, res, delta, iters=50):
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
    src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
    vm = gpa.GaussianVoxelMapGPU(res, target_points_drop_rate=0.0)
    vm.insert(tgt)
    f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
    om = oracle.OracleVoxelMap(res)
    om.insert(d["target_points"], d["target_covs"])
    Lo = oracle.OracleVGICPFactor(om, d["source_points"], d["source_covs"], oracle.max_threads()).linearize(delta)
    pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
    rows = []
    for v in variants:
        _capi.check(lib.gp_debug_set_variant(v), "variant")
        arr = (C.c_void_p * 1)(f._h.value)
        batch = C.c_void_p()
        s = C.c_void_p()
        lib.gp_stream_create(C.byref(s))
        _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
        out = np.zeros((1, 122))
        _capi.check(lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data), "lin")
        L = gpa.LinearizedSystem6.from_doubles(out[0])
        errs = {k: float(np.linalg.norm(getattr(L, k) - getattr(Lo, k)) / np.linalg.norm(getattr(Lo, k))) for k in BLOCKS}
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        best = (1e9, 0, 0)
        for _ in range(3):
            _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, iters, C.byref(a), C.byref(b), C.byref(c)), "time")
            best = min(best, (b.value, a.value, c.value))
        t0 = time.perf_counter()
        for _ in range(200):
            lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
        wall = (time.perf_counter() - t0) / 200 * 1e3
        alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
        rows.append(dict(case=name, variant=v, tile_ms=round(best[0], 5), pass_ms=round(best[1], 5), fin_ms=round(best[2], 5), sync_call_ms=round(wall, 5),
                         frac=round(alg / (best[0] * 1e-3) / 8e12, 4), max_rel_err=max(errs.values()), inliers_ok=L.num_inliers == Lo.num_inliers))
        print(json.dumps(rows[-1]), flush=True)
        lib.gp_vgicp_batch_destroy(batch)
        lib.gp_stream_destroy(s)
    return rows


d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
run_case("c2_1M", d, 0.5, delta)
k = np.load(os.path.join(ROOT, "tests/golden/kitti00_dec8.npz"))
run_case("kitti00_dec8", {n: k[n] for n in k.files}, 0.5, synthetic.expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03]), iters=200)
d2 = synthetic.make_pair(120000, 120000, seed=3)
run_case("pair_120k", d2, 0.5, d2["T_true"] @ synthetic.expmap([0.002, -0.001, 0.0015, 0.02, -0.01, 0.015]), iters=100)
