"""Workloads for profiling the kernels next to the headline: the voxel-map build (2 M points, 0.5 m) and BASELINE configs[4]
(k-NN covariances + GICP linearise at 1 M points).  Run under rocprofv3 --kernel-trace --stats (scripts/r02_evidence.sh).
With `counters` the binned search counts its own work (queries, f32/f64 distance evaluations, block entries, cells) instead.
Usage: python scripts/r02_profile_aux.py [map|c5|counters] [iters]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import synthetic  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "map"
if os.environ.get("GP_KNN_MODE"):
    gpa.load().gp_debug_set_knn_structure(int(os.environ["GP_KNN_MODE"]))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = synthetic.make_c2_workload(1_000_000, 2_000_000 if what == "map" else 1_000_000, seed=42)
if what == "map":
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
        t = time.perf_counter()
        vm.insert(tgt)
        ts.append(time.perf_counter() - t)
    print(f"voxel map build, 2M points @0.5 m: {vm.voxelmap_info.num_voxels} voxels, wall median {np.median(ts)*1e3:.3f} ms, min {min(ts)*1e3:.3f} ms", flush=True)
elif what == "counters":
    import ctypes as C
    import json
    lib = gpa.load()
    tgt, src = gpa.PointCloudGPU(d["target_points"]), gpa.PointCloudGPU(d["source_points"])
    names = ["shell_walks", "f32_distances", "f64_distances", "block_entries", "cells", "octant_stages"]

    def read():
        out = (C.c_ulonglong * 6)()
        lib.gp_debug_knn_counters(0, out)
        return dict(zip(names, [int(v) for v in out]))

    lib.gp_debug_knn_counters(1, None)
    gpa.estimate_covariances_gpu(src, 10)
    c = read()
    c.update(kernel="covariance_kernel<10>", queries=src.size())
    print(json.dumps(c), flush=True)
    gpa.estimate_covariances_gpu(tgt, 10)
    lib.gp_debug_knn_counters(1, None)
    f = gpa.IntegratedGICPFactorGPU(0, 1, tgt, src)
    delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    L = f.linearize_delta(delta)
    c = read()
    c.update(kernel="gicp_tile_kernel<LIN>", queries=src.size(), inliers=int(L.num_inliers))
    print(json.dumps(c), flush=True)
else:
    tgt, src = gpa.PointCloudGPU(d["target_points"]), gpa.PointCloudGPU(d["source_points"])
    torch.cuda.synchronize()
    tc = []
    for _ in range(iters):
        t = time.perf_counter()
        gpa.estimate_covariances_gpu(src, 10)
        tc.append(time.perf_counter() - t)
    gpa.estimate_covariances_gpu(tgt, 10)
    f = gpa.IntegratedGICPFactorGPU(0, 1, tgt, src)
    delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    tg = []
    for _ in range(iters):
        t = time.perf_counter()
        L = f.linearize_delta(delta)
        tg.append(time.perf_counter() - t)
    print(f"C5 1M: covariances wall median {np.median(tc)*1e3:.3f} ms, GICP linearise wall median {np.median(tg)*1e3:.3f} ms, inliers {L.num_inliers}", flush=True)
