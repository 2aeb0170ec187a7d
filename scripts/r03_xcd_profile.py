"""Per-XCD profile of the tile kernel inside synchronous (fused) steps: when the workgroups of each XCD start, when its last row is stored, over `reps` traced steps.
Also with per-XCD share weights given as 8 comma-separated permille values (GP_TUNE_XCD_WEIGHT_x).  One JSON object per configuration.
Usage: python scripts/r03_xcd_profile.py [reps=12] [weights;weights;...]"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
configs = [None] + ([[int(x) for x in c.split(",")] for c in sys.argv[2].split(";")] if len(sys.argv) > 2 else [])
lib = gpa.load()
d = synthetic.make_c2_workload()
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
pose = np.ascontiguousarray((d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])).T).reshape(1, 16).copy()
out = np.zeros((1, 122))
for weights in configs:
    f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
    arr = (C.c_void_p * 1)(f._h.value)
    batch, s = C.c_void_p(), C.c_void_p()
    lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
    if weights:
        for x, w in enumerate(weights):
            _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 8 + x, w), "xcd weight")
    for _ in range(50):
        lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
    ts = np.empty(1500)
    for i in range(len(ts)):
        t0 = time.perf_counter_ns()
        lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
        ts[i] = time.perf_counter_ns() - t0
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 50, C.byref(a), C.byref(b), C.byref(c)), "time")
    starts, ends, alls = [], [], []
    for rep in range(reps):
        trace = torch.zeros((2048, 16), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        _capi.check(lib.gp_vgicp_batch_set_trace_buffer(batch, C.c_void_p(trace.data_ptr())), "trace")
        for _ in range(3):
            lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)  # the third step's stamps stay
        torch.cuda.synchronize()
        raw = trace.cpu().numpy()[:2047]
        raw = raw[raw[:, 10] > 0].astype(np.float64)
        t0 = raw[:, 10].min()
        xcc = raw[:, 9].astype(int) & 15
        starts.append([float(np.median(raw[xcc == x, 10] - t0)) / 100 for x in range(8)])
        ends.append([float((raw[xcc == x, 12] - t0).max()) / 100 for x in range(8)])
        alls.append(float((raw[:, 15].max() - t0)) / 100)
        lib.gp_vgicp_batch_set_trace_buffer(batch, None)
    starts, ends = np.array(starts), np.array(ends)
    print(json.dumps(dict(xcd_weights=weights, call_us_median=round(float(np.median(ts)) / 1e3, 2), tile_us_back_to_back=round(b.value * 1e3, 2),
                          start_median_us_by_xcd=[round(x, 2) for x in np.median(starts, axis=0)], last_row_us_by_xcd_median=[round(x, 2) for x in np.median(ends, axis=0)],
                          last_row_us_by_xcd_p90=[round(x, 2) for x in np.percentile(ends, 90, axis=0)], words_out_us_median=round(float(np.median(alls)), 2))), flush=True)
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)
