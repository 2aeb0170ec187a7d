"""C4 shard (512 factors x 32768 points, the shard of one GPU of eight: factor list [0:512] of synthetic.c4_factor_pairs) under different schedules, for the
traffic reconciliation of VERDICT r02 #5.  Configurations "interleave:policy:tile_chunks" (GP_TUNE_TILE_INTERLEAVE : GP_TUNE_SOURCE_POLICY :
GP_TUNE_TILE_CHUNKS).  Without PMC=1: tile-kernel time per configuration (HIP events, gp_vgicp_batch_time_linearize).  With PMC=1 (under
rocprofv3 --pmc ...): exactly REPS synchronous linearise calls per configuration, in order, nothing else -- the counter rows of vgicp_stream_kernel are
then grouped by dispatch order (scripts/r03_c4_traffic_parse.py)."""
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
from gtsam_points_amd import _capi, synthetic  # noqa: E402

CONFIGS = os.environ.get("C4_CONFIGS", "0:0:0,1:0:0,0:2:0,0:0:8,1:0:8,0:0:16").split(",")
REPS = 4
PMC = bool(os.environ.get("PMC"))
lib = gpa.load()
WORKLOAD = os.environ.get("WORKLOAD", "c4")  # c4: the shard of one GPU of eight; c4all: all 4096 factors; c3: the 256-factor submap graph
if WORKLOAD == "c3":
    g = synthetic.make_c3_graph()
    pairs = g["pairs"]
    sub = {i: (p, c, None) for i, (p, c) in enumerate(g["clouds"])}
    deltas = g["deltas"]
else:
    pairs = synthetic.c4_factor_pairs() if WORKLOAD == "c4all" else synthetic.c4_factor_pairs()[:512]
    sub = synthetic.make_c4_submaps(sorted({i for p in pairs for i in p}))
    deltas = [synthetic.c4_delta(sub, t, s_) for t, s_ in pairs]
need = sorted({i for p in pairs for i in p})
clouds = {i: gpa.PointCloudGPU(sub[i][0], sub[i][1]) for i in need}
maps = {}
for t in sorted({t for t, _ in pairs}):
    m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
    m.insert(clouds[t])
    maps[t] = m
factors = [gpa.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s]) for t, s in pairs]
F = len(factors)
arr = (C.c_void_p * F)(*[f._h.value for f in factors])
batch, s = C.c_void_p(), C.c_void_p()
lib.gp_stream_create(C.byref(s))
_capi.check(lib.gp_vgicp_batch_create(arr, F, s, C.byref(batch)), "batch")
poses = np.stack([np.ascontiguousarray(np.asarray(d_).T).reshape(16) for d_ in deltas]).copy()
out = np.zeros((F, 122))
unique_bytes = sum(48 * len(sub[i][0]) for i in {s_ for _, s_ in pairs}) + sum(64 * maps[t].voxelmap_info.num_voxels for t in maps)
torch.cuda.synchronize()
for cfg in CONFIGS:
    il, pol, tc = (int(x) for x in cfg.split(":"))
    _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_TILE_INTERLEAVE, il), "interleave")
    _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_SOURCE_POLICY, pol), "policy")
    _capi.check(lib.gp_vgicp_batch_set_tuning(batch, _capi.GP_TUNE_TILE_CHUNKS, tc), "tile chunks")
    for _ in range(REPS):
        _capi.check(lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out.ctypes.data), "linearize")
    if not PMC:
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        best = 1e9
        for _ in range(3):
            _capi.check(lib.gp_vgicp_batch_time_linearize(batch, poses.ctypes.data, 20, C.byref(a), C.byref(b), C.byref(c)), "time")
            best = min(best, b.value)
        alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
        t0 = time.perf_counter()
        for _ in range(20):
            lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out.ctypes.data)
        wall = (time.perf_counter() - t0) / 20 * 1e3
        print(json.dumps(dict(workload=WORKLOAD, sync_call_ms=round(wall, 4), config=cfg, interleave=il, policy=pol, tile_chunks=tc, tile_ms=round(best, 5), algorithmic_bytes=alg, unique_bytes=unique_bytes,
                              frac_algorithmic=round(alg / (best * 1e-3) / 8e12, 4), inliers=float(out[:, 0].sum()))), flush=True)
print(json.dumps(dict(configs=CONFIGS, reps=REPS, pmc=PMC)), flush=True)
