"""Randomised soak of the round-3 synchronous paths: a large single factor (fused finalize by parts), a small single factor and a 12-factor batch (fused by factor),
driven with random poses while the knobs change under them -- fused on / off, plan balance and tile size (table rebuilds), timing mode, the arrival-skew test hook --
from two threads at once (one per group of batches).  Every fused record is compared, bit for bit, with the two-kernel form of the same call at the same pose, every
error evaluation likewise.  Prints one summary line; exits non-zero on the first mismatch.  Usage: python scripts/r03_fuzz.py [seconds=20] [seed=1]"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import _capi, synthetic  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = gpa.load()
big = synthetic.make_c2_workload(600_000, 400_000, seed=3)
small = synthetic.make_c2_workload(30_000, 60_000, seed=4)


def build(d):
    tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"])
    src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
    vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    return tgt, src, vm


def make_batch(factors):
    arr = (C.c_void_p * len(factors))(*[f._h.value for f in factors])
    batch, s = C.c_void_p(), C.c_void_p()
    _capi.check(lib.gp_stream_create(C.byref(s)), "stream")
    _capi.check(lib.gp_vgicp_batch_create(arr, len(factors), s, C.byref(batch)), "batch")
    return batch, s


keep = []
tb, sb, vb = build(big)
ts, ss, vs = build(small)
keep += [tb, sb, vb, ts, ss, vs]
f_big = gpa.IntegratedVGICPFactorGPU(0, 1, vb, sb)
f_small = gpa.IntegratedVGICPFactorGPU(0, 1, vs, ss)
subs = []
for k in range(4):
    dd = synthetic.make_c2_workload(20_000 + 3_000 * k, 50_000, seed=10 + k)
    subs.append((dd, *build(dd)))
f_many = [gpa.IntegratedVGICPFactorGPU(0, 1, subs[i % 4][3], subs[(i + 1) % 4][2]) for i in range(12)]
groups = [  # (name, batch, true poses per factor, tunable keys)
    ("large single factor", *make_batch([f_big]), [big["T_true"]]),
    ("small single factor", *make_batch([f_small]), [small["T_true"]]),
    ("12-factor batch", *make_batch(f_many), [subs[(i + 1) % 4][0]["T_true"] for i in range(12)]),
]
stats = {g[0]: dict(calls=0, errors=0, rebuilds=0, skews=0) for g in groups}
failed = []
deadline = time.time() + seconds


def drive(name, batch, stream, truths, rng):
    F = len(truths)
    out_a, out_b = np.zeros((F, 122)), np.zeros((F, 122))
    ea, eb = np.zeros(F), np.zeros(F)
    st = stats[name]
    while time.time() < deadline and not failed:
        P = np.ascontiguousarray(np.stack([(T @ synthetic.expmap(rng.uniform(-2e-3, 2e-3, 6))).T.reshape(16) for T in truths]))
        P2 = np.ascontiguousarray(np.stack([(T @ synthetic.expmap(rng.uniform(-2e-3, 2e-3, 6))).T.reshape(16) for T in truths]))
        r = rng.random()
        if r < 0.08:
            _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 5, int(rng.choice([0, 100, 250, -1]))), "balance")  # rebuilds the table
            st["rebuilds"] += 1
        elif r < 0.14:
            _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 18, int(rng.choice([0, 1, 2, 4, 8]))), "tile chunks")
            st["rebuilds"] += 1
        elif r < 0.18:
            _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 20, int(rng.integers(1, 9))), "arrival skew")  # the next fused step has to recover
            st["skews"] += 1
        elif r < 0.22:
            _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 7, int(rng.integers(0, 2))), "timing")
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 17, 1), "fused")
        _capi.check(lib.gp_vgicp_batch_linearize(batch, P.ctypes.data, out_a.ctypes.data), "linearize fused")
        if rng.random() < 0.5:
            _capi.check(lib.gp_vgicp_batch_compute_error(batch, P.ctypes.data, P2.ctypes.data, ea.ctypes.data), "error fused")
        else:
            ea[:] = np.nan
        _capi.check(lib.gp_vgicp_batch_set_tuning(batch, 17, 0), "two-kernel")
        _capi.check(lib.gp_vgicp_batch_linearize(batch, P.ctypes.data, out_b.ctypes.data), "linearize two-kernel")
        if not np.array_equal(out_a, out_b):
            failed.append(f"{name}: records differ after {st['calls']} calls")
            return
        if not np.isnan(ea[0]):
            _capi.check(lib.gp_vgicp_batch_compute_error(batch, P.ctypes.data, P2.ctypes.data, eb.ctypes.data), "error two-kernel")
            if not np.array_equal(ea, eb):
                failed.append(f"{name}: errors differ after {st['calls']} calls")
                return
            st["errors"] += 1
        if out_a[0, 0] < 100:
            failed.append(f"{name}: implausible record (inliers {out_a[0, 0]})")
            return
        st["calls"] += 1


threads = [threading.Thread(target=drive, args=(g[0], g[1], g[2], g[3], np.random.default_rng(seed + 17 * i))) for i, g in enumerate(groups)]
for t in threads:
    t.start()
for t in threads:
    t.join()
for g in groups:
    lib.gp_vgicp_batch_destroy(g[1])
    lib.gp_stream_destroy(g[2])
print({k: v for k, v in stats.items()}, "FAILED: " + "; ".join(failed) if failed else "FUZZ_OK", flush=True)
sys.exit(1 if failed else 0)
