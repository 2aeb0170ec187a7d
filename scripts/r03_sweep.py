"""Round-3 tuning sweep of the tile kernel on the GPU box (one JSON object per line on stdout), all through per-batch tuning
(gp_vgicp_batch_set_tuning: nothing process-global):
  * cases "family:policy:balance" (GP_KERNEL_* : source policy 0 per batch / 1 default / 2 non-temporal : stream plan 1 late-light / 0 flat)
    on the C2 workload: tile-kernel time (HIP events, best of 3 x 50 launches), whole device pass, synchronous call, parity vs the oracle
  * per-workgroup timeline of the traced build incl. the hardware placement (HW_ID) of every workgroup and the chunks every CU carried
  * the real kernel on an 8 M-point source (working set > the 256 MiB Infinity Cache)
Usage: python scripts/r03_sweep.py [cases, e.g. 8:0:0,12:0:1,12:0:0] [--big] [--no-trace]"""
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
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cases = [tuple(int(x) for x in c.split(":")) for c in (args[0] if len(args) > 0 else "8:0:0,12:0:0,12:0:100,12:0:200").split(",")]
BIG = "--big" in sys.argv
TRACE = "--no-trace" not in sys.argv
KERNEL, POLICY, BALANCE = 0, 1, 5  # GP_TUNE_*
XCD_WEIGHTS = {0: [1000] * 8, 2: [1045, 1045, 1045, 1045, 940, 925, 975, 980], 3: [1015, 1015, 1015, 1015, 980, 975, 990, 995]}
BLOCKS = ["H_target", "H_source", "H_target_source", "b_target", "b_source"]


def make_batch(f, case):
    arr = (C.c_void_p * 1)(f._h.value)
    batch, s = C.c_void_p(), C.c_void_p()
    lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
    fam, pol, bal = case[:3]
    _capi.check(lib.gp_vgicp_batch_set_tuning(batch, KERNEL, fam), "kernel")
    _capi.check(lib.gp_vgicp_batch_set_tuning(batch, POLICY, pol), "policy")
    _capi.check(lib.gp_vgicp_batch_set_tuning(batch, BALANCE, bal), "balance")
    if len(case) > 4:
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 17, case[4]), "overlap")  # GP_TUNE_FUSED_FINALIZE
    if len(case) > 5:
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 19, case[5]), "max workgroups")  # GP_TUNE_MAX_WORKGROUPS
    wsel = case[3] if len(case) > 3 else 1  # XCD weights: 0 equal shares, 1 the library's table (default), 2.. alternatives
    if wsel != 1:
        for x, w in enumerate(XCD_WEIGHTS[wsel]):
            _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 8 + x, w), "xcd weight")
    return batch, s


def time_batch(batch, pose, iters=50, reps=3):
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    best = (1e9, 0, 0)
    for _ in range(reps):
        _capi.check(lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, iters, C.byref(a), C.byref(b), C.byref(c)), "time")
        best = min(best, (b.value, a.value, c.value))
    return best


def run_case(name, d, res, delta, Lo, iters=50):
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
    src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
    vm = gpa.GaussianVoxelMapGPU(res, target_points_drop_rate=0.0)
    t0 = time.perf_counter()
    vm.insert(tgt)
    t_map = time.perf_counter() - t0
    f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
    pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
    out = np.zeros((1, 122))
    for case in cases:
        batch, s = make_batch(f, case)
        _capi.check(lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data), "lin")
        L = gpa.LinearizedSystem6.from_doubles(out[0])
        errs = {k: float(np.linalg.norm(getattr(L, k) - getattr(Lo, k)) / np.linalg.norm(getattr(Lo, k))) for k in BLOCKS} if Lo is not None else {}
        best = time_batch(batch, pose, iters)
        t0 = time.perf_counter()
        for _ in range(200):
            lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
        wall = (time.perf_counter() - t0) / 200 * 1e3
        eout = np.zeros(1)
        t0 = time.perf_counter()
        for _ in range(200):
            lib.gp_vgicp_batch_compute_error(batch, pose.ctypes.data, pose.ctypes.data, eout.ctypes.data)
        err_wall = (time.perf_counter() - t0) / 200 * 1e3
        alg = int(lib.gp_vgicp_batch_algorithmic_bytes(batch))
        eff = C.c_int(-2)
        lib.gp_vgicp_batch_get_tuning(batch, 6, C.byref(eff))
        print(json.dumps(dict(case=name, family=case[0], policy=case[1], balance=case[2], xcd_weights=(case[3] if len(case) > 3 else 1), overlap=(case[4] if len(case) > 4 else 0), max_wgs=(case[5] if len(case) > 5 else 1024), effective_family=eff.value, tile_ms=round(best[0], 5), pass_ms=round(best[1], 5),
                              fin_ms=round(best[2], 5), sync_call_ms=round(wall, 5), error_sync_call_ms=round(err_wall, 5),
                              frac=round(alg / (best[0] * 1e-3) / 8e12, 4), alg_bytes=alg, max_rel_err=max(errs.values()) if errs else None,
                              inliers_ok=(L.num_inliers == Lo.num_inliers) if Lo is not None else None, has_grid=int(lib.gp_voxelmap_has_block_grid(vm._h)),
                              voxels=vm.voxelmap_info.num_voxels, map_build_ms=round(t_map * 1e3, 3))), flush=True)
        lib.gp_vgicp_batch_destroy(batch)
        lib.gp_stream_destroy(s)
    return f, vm, src, tgt


def trace_case(f, delta, label, case):
    stagger = 0
    batch, s = make_batch(f, case)
    pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
    out = np.zeros((1, 122))
    for _ in range(5):
        lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
    trace = torch.zeros((2048, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    _capi.check(lib.gp_vgicp_batch_set_trace_buffer(batch, C.c_void_p(trace.data_ptr())), "trace")
    lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data)
    torch.cuda.synchronize()
    lib.gp_vgicp_batch_set_trace_buffer(batch, None)
    raw = trace.cpu().numpy()[:2047]  # (row 2047 belongs to the finalize kernel: scripts/trace_finalize.py)
    raw = raw[raw[:, 0] > 0]
    # rows stamped by an earlier launch (a workgroup whose tile index is not used by the traced launch keeps its old stamps)
    t = raw[:, :8].astype(np.float64)
    # one time axis for the whole device: the 100 MHz constant clock (s_memrealtime), 10 ns per tick
    rs, re_ = raw[:, 10].astype(np.float64) / 100.0, raw[:, 11].astype(np.float64) / 100.0
    ok = np.abs(rs - np.median(rs)) < 100.0  # (a row of a tile index the traced launch did not use keeps older stamps)
    rs, re_ = rs[ok] - rs[ok].min(), re_[ok] - rs[ok].min()
    device_axis = dict(wgs=int(ok.sum()), start_p50_us=round(float(np.median(rs)), 2), start_p90_us=round(float(np.percentile(rs, 90)), 2), start_max_us=round(float(rs.max()), 2),
                       end_min_us=round(float(re_.min()), 2), end_p10_us=round(float(np.percentile(re_, 10)), 2), end_p50_us=round(float(np.median(re_)), 2),
                       end_p90_us=round(float(np.percentile(re_, 90)), 2), end_max_us=round(float(re_.max()), 2), life_p50_us=round(float(np.median(re_ - rs)), 2))
    hw = raw[:, 8]
    xcc = raw[:, 9] & 0xF
    wave_slot, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    names = ["start", "chunk0", "look0", "step0_done", "look1", "step1_done", "steps_done", "end"]
    dur = np.diff(t, axis=1) / 2100.0
    # per clock domain (XCC): lifetimes and end skew
    doms = []
    for x in sorted(set(xcc.tolist())):
        r = t[xcc == x]
        s0 = r[:, 0].min()
        life = (r[:, 7] - r[:, 0]) / 2100
        doms.append(dict(xcc=int(x), wgs=int(len(r)), start_skew_us=round(float((r[:, 0].max() - s0) / 2100), 2), start_p50_us=round(float(np.median(r[:, 0] - s0) / 2100), 2),
                         first_end_us=round(float((r[:, 7].min() - s0) / 2100), 2), last_end_us=round(float((r[:, 7].max() - s0) / 2100), 2),
                         median_life_us=round(float(np.median(life)), 2), p90_life_us=round(float(np.percentile(life, 90)), 2), max_life_us=round(float(life.max()), 2)))
    # placement: workgroups per (xcc, se, sh, cu), and which tile indices share a CU
    keys = xcc * 4096 + se * 512 + sh * 256 + cu
    uniq, counts = np.unique(keys, return_counts=True)
    # end of the last workgroup of every CU on the device-wide axis, by how many workgroups the CU got, and per XCC
    re_all = raw[:, 11].astype(np.float64) / 100.0 - (raw[:, 10].astype(np.float64) / 100.0)[ok].min()
    cu_end = {}
    for k_, e_, good in zip(keys.tolist(), re_all.tolist(), ok.tolist()):
        if good:
            cu_end[k_] = max(cu_end.get(k_, 0.0), e_)
    cnt = dict(zip(uniq.tolist(), counts.tolist()))
    by_load = {}
    for k_, e_ in cu_end.items():
        by_load.setdefault(cnt[k_], []).append(e_)
    device_axis["cu_last_end_by_wgs_per_cu"] = {str(n): dict(cus=len(v), p50=round(float(np.median(v)), 2), p90=round(float(np.percentile(v, 90)), 2), max=round(float(max(v)), 2)) for n, v in sorted(by_load.items())}
    by_xcc = {}
    for k_, e_ in cu_end.items():
        by_xcc.setdefault(k_ // 4096, []).append(e_)
    device_axis["cu_last_end_by_xcc"] = {str(x): dict(p50=round(float(np.median(v)), 2), max=round(float(max(v)), 2)) for x, v in sorted(by_xcc.items())}
    device_axis["wg_start_p50_by_xcc"] = {str(int(x)): round(float(np.median(rs[xcc[ok] == x])), 2) for x in sorted(set(xcc[ok].tolist()))}
    tile_ids = np.nonzero(trace.cpu().numpy()[:2047, 0] > 0)[0]
    same_cu = {}
    for k_, tid in zip(keys.tolist(), tile_ids.tolist()):
        same_cu.setdefault(k_, []).append(tid)
    example = [v for v in same_cu.values()][:4]
    print(json.dumps(dict(trace=label, stagger=stagger, wgs=int(len(t)), phases=names,
                          phase_median_us=[round(float(np.median(dur[:, k])), 3) for k in range(7)], phase_p90_us=[round(float(np.percentile(dur[:, k], 90)), 3) for k in range(7)],
                          device_axis=device_axis, domains=doms, cus_used=int(len(uniq)), wgs_per_cu_hist=np.bincount(counts).tolist(), wave_slot_hist=np.bincount(wave_slot.astype(np.int64), minlength=16).tolist(),
                          simd_hist=np.bincount(simd.astype(np.int64), minlength=4).tolist(), tiles_sharing_a_cu_examples=example)), flush=True)
    # start time of a workgroup against its position in the dispatch order (row = tile index; stream family: x * 128 + q, q = position in the XCD's share)
    tid = tile_ids[ok] if len(tile_ids) == len(ok) else None
    if tid is not None and case[0] == 12:
        q = tid % 128
        rounds = {}
        for r_ in range(4):
            m = (q >= 32 * r_) & (q < 32 * (r_ + 1))
            if m.any():
                rounds[str(r_)] = dict(start_p50=round(float(np.median(rs[m])), 2), end_p50=round(float(np.median(re_[m])), 2), end_max=round(float(re_[m].max()), 2), wgs=int(m.sum()))
        print(json.dumps(dict(trace=label, by_dispatch_round_of_32_per_xcd=rounds)), flush=True)
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)


d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
om = oracle.OracleVoxelMap(0.5)
om.insert(d["target_points"], d["target_covs"])
Lo = oracle.OracleVGICPFactor(om, d["source_points"], d["source_covs"], oracle.max_threads()).linearize(delta)
f, vm, src, tgt = run_case("c2_1M", d, 0.5, delta, Lo)
if TRACE:
    for case in [c for c in cases if c[0] == 12]:
        trace_case(f, delta, f"c2_1M family {case[0]} policy {case[1]} balance {case[2]} weights {case[3] if len(case) > 3 else 1}", case)

k = np.load(os.path.join(ROOT, "tests/golden/kitti00_dec8.npz"))
dk = {n: k[n] for n in k.files}
dlt = synthetic.expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
omk = oracle.OracleVoxelMap(0.5)
omk.insert(dk["target_points"], dk["target_covs"])
Lok = oracle.OracleVGICPFactor(omk, dk["source_points"], dk["source_covs"], 4).linearize(dlt)
run_case("kitti00_dec8", dk, 0.5, dlt, Lok, iters=200)

if BIG:
    # the real kernel on a working set beyond the 256 MiB Infinity Cache: 8 M source points (384 MB) vs the same 2 M-point map
    big = synthetic.make_c2_workload(8_000_000, 2_000_000, seed=42)
    run_case("c2_8M_source", big, 0.5, big["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015]), None, iters=20)
