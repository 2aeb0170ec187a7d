"""Time + parity-check selected tile-kernel variants on the C2 workload: python scripts/variant_test.py 0 1 2"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
from gtsam_points_amd import _capi, synthetic
lib = gpa.load()
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6]
d = synthetic.make_c2_workload()
delta = d["T_true"] @ synthetic.expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
tgt = gpa.PointCloudGPU(d["target_points"], d["target_covs"]); src = gpa.PointCloudGPU(d["source_points"], d["source_covs"])
vm = gpa.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0); vm.insert(tgt)
f = gpa.IntegratedVGICPFactorGPU(0, 1, vm, src)
pose = np.ascontiguousarray(delta.T).reshape(1, 16).copy()
ref = None
for variant in variants:
    lib.gp_debug_set_variant(variant)
    arr = (C.c_void_p * 1)(f._h.value); batch, s = C.c_void_p(), C.c_void_p(); lib.gp_stream_create(C.byref(s))
    _capi.check(lib.gp_vgicp_batch_create(arr, 1, s, C.byref(batch)), "batch")
    out = np.zeros(122); _capi.check(lib.gp_vgicp_batch_linearize(batch, pose.ctypes.data, out.ctypes.data), "lin")
    if ref is None: ref = out.copy()
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    a, b, c = C.c_float(), C.c_float(), C.c_float(); best = (1e9, 0, 0)
    for _ in range(3):
        lib.gp_vgicp_batch_time_linearize(batch, pose.ctypes.data, 50, C.byref(a), C.byref(b), C.byref(c))
        if b.value < best[0]: best = (b.value, a.value, c.value)
    print(f"variant {variant:3d}: pass {best[1]*1e3:.2f} us tile {best[0]*1e3:.2f} us fin {best[2]*1e3:.2f} us  inl {out[0]:.0f} err {out[1]:.6f} "
          f"relH {rel(out[2:110], ref[2:110]):.2e} relb {rel(out[110:122], ref[110:122]):.2e}", flush=True)
    lib.gp_vgicp_batch_destroy(batch)
