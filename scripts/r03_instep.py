"""Why is the tile kernel 1.1-1.3 us slower inside a synchronous step than back to back?  Per-workgroup timelines (traced build) of the SAME kernel
in both launch patterns: (a) in step -- one synchronous gp_vgicp_batch_linearize after others; (b) back to back -- the last launch of the tile-only loop
of gp_vgicp_batch_time_linearize.  Printed per pattern: workgroup start spread, phase medians, end times, and the SHADER CLOCK seen by the kernel
(s_memtime ticks per s_memrealtime tick: the constant clock runs at 100 MHz)."""
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

lib = gpa.load()
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
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
out = np.zeros((1, 122))
for _ in range(20):
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)


def analyse(raw, label):
    raw = raw[:2047]
    raw = raw[raw[:, 0] > 0]
    t = raw[:, :8].astype(np.float64)
    rs, re_ = raw[:, 10].astype(np.float64), raw[:, 11].astype(np.float64)  # 100 MHz ticks
    ok = np.abs(rs - np.median(rs)) < 10000
    t, rs, re_ = t[ok], rs[ok], re_[ok]
    ticks = t[:, 7] - t[:, 0]
    real = (re_ - rs) * 10.0  # ns
    mhz = float(np.median(ticks / real * 1e3))
    s0 = rs.min()
    res = dict(pattern=label, wgs=int(len(t)), shader_clock_mhz_median=round(mhz, 1), shader_clock_mhz_p10=round(float(np.percentile(ticks / real * 1e3, 10)), 1),
               start_p50_us=round(float(np.median(rs - s0)) / 100, 2), start_max_us=round(float((rs - s0).max()) / 100, 2),
               end_p50_us=round(float(np.median(re_ - s0)) / 100, 2), end_p90_us=round(float(np.percentile(re_ - s0, 90)) / 100, 2), end_max_us=round(float((re_ - s0).max()) / 100, 2),
               life_p50_us=round(float(np.median(re_ - rs)) / 100, 2),
               phase_median_us_at_measured_clock=[round(float(np.median(np.diff(t, axis=1)[:, k])) / mhz, 3) for k in range(7)])
    print(json.dumps(res), flush=True)


for rep in range(3):
    trace = torch.zeros((2048, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    _capi.check(lib.gp_vgicp_batch_set_trace_buffer(batch, C.c_void_p(trace.data_ptr())), "trace")
    time.sleep(0.0001)
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)  # in step: behind an idle queue
    torch.cuda.synchronize()
    analyse(trace.cpu().numpy(), "in_step")
    trace.zero_()
    torch.cuda.synchronize()
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 6, C.byref(a), C.byref(b), C.byref(c)), "time")  # the last tile launch: back to back
    torch.cuda.synchronize()
    analyse(trace.cpu().numpy(), "back_to_back")
    lib.gp_vgicp_batch_set_trace_buffer(batch, None)
