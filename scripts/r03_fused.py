"""Fused finalize (GP_TUNE_FUSED_FINALIZE = 1: the last tile workgroup of each eighth of the tile list sums its rows and hands them to the
host) against the two-kernel form, on the C2 workload: wall time of the synchronous gp_vgicp_batch_linearize call (median / mean / p90 over
`steps` calls, the modes alternating in blocks), and that the records are the same bits.  One JSON object per block on stdout.
Usage: python scripts/r03_fused.py [steps=2000] [blocks=6]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lib = gpa.load()
for name, n_src in [("c2_1M", 1_000_000), ("c2_300k", 300_000), ("c2_4M", 4_000_000)]:
    d = synthetic.make_c2_workload(n_src, 500_000, seed=1) if n_src != 1_000_000 else synthetic.make_c2_workload()
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
    src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
    arr = (C.c_void_p * 1)(f._h.value)
    batch, s = C.c_void_p(), C.c_void_p()
    lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
    pose = np.ascontiguousarray((d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])).T).reshape(1, 16).copy()
    out = np.zeros((1, 122))
    records = {}
    for blk in range(blocks):
        mode = blk % 2
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 17, mode), "fused")
        for _ in range(50):
            _capi.check(lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data), "linearize")
        ts = np.empty(steps)
        for i in range(steps):
            t0 = time.perf_counter_ns()
            lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
            ts[i] = time.perf_counter_ns() - t0
        te = np.empty(steps)
        e = C.c_double()
        for i in range(steps):
            t0 = time.perf_counter_ns()
            lib.gp_vgicp_batch_compute_error(batch, pose.ctypes.data, pose.ctypes.data, C.byref(e))
            te[i] = time.perf_counter_ns() - t0
        records.setdefault(mode, out.copy())
        same = bool(np.array_equal(records[mode], out) and np.array_equal(records[0], out))
        print(json.dumps(dict(case=name, fused=mode, steps=steps, call_us_median=round(float(np.median(ts)) / 1e3, 2), call_us_mean=round(float(ts.mean()) / 1e3, 2),
                              call_us_p90=round(float(np.percentile(ts, 90)) / 1e3, 2), error_call_us_median=round(float(np.median(te)) / 1e3, 2), error=e.value,
                              record_equals_two_kernel_form=same)), flush=True)
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)

# per-workgroup timeline of the fused tail (traced build of the stream kernel; 100 MHz constant clock): slot 10 start, 12 row stored and acknowledged,
# 13 arrival atomic returned, 14 (last workgroup of a part) rows summed, 15 sums and completion word on their way to the host, 11 end
d = synthetic.make_c2_workload()
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
arr = (C.c_void_p * 1)(f._h.value)
batch, s = C.c_void_p(), C.c_void_p()
lib.gp_stream_create(C.byref(s))
_capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
_capi.check(lib.gp_vgicp_batch_set_tuning(batch, 17, 1), "fused")
pose = np.ascontiguousarray((d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])).T).reshape(1, 16).copy()
out = np.zeros((1, 122))
for _ in range(20):
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
for rep in range(3):
    trace = torch.zeros((2048, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    _capi.check(lib.gp_vgicp_batch_set_trace_buffer(batch, C.c_void_p(trace.data_ptr())), "trace")
    time.sleep(0.0001)
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
    torch.cuda.synchronize()
    raw = trace.cpu().numpy()[:2047]
    raw = raw[raw[:, 10] > 0].astype(np.float64)
    t0 = raw[:, 10].min()
    us = lambda x: round(float(x) / 100, 2)
    last = raw[raw[:, 15] > 0]
    print(json.dumps(dict(trace="fused_tail", wgs=int(len(raw)), start_max_us=us((raw[:, 10] - t0).max()), row_stored_p50_us=us(np.median(raw[:, 12] - t0)),
                          row_stored_max_us=us((raw[:, 12] - t0).max()), atomic_us_p50=us(np.median(raw[:, 13] - raw[:, 12])), atomic_us_max=us((raw[:, 13] - raw[:, 12]).max()),
                          arrived_max_us=us((raw[:, 13] - t0).max()), finalizers=int(len(last)),
                          finalizer_arrived_us=[us(x - t0) for x in last[:, 13]], rows_summed_after_us=[us(x) for x in last[:, 14] - last[:, 13]],
                          host_store_after_us=[us(x) for x in last[:, 15] - last[:, 14]], all_words_out_us=us((last[:, 15] - t0).max()) if len(last) else None,
                          end_max_us=us((raw[:, 11] - t0).max()))), flush=True)
    lib.gp_vgicp_batch_set_trace_buffer(batch, None)
