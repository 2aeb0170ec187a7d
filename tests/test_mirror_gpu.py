"""Packed source mirrors (round 4): the stream kernels read a private 36-B-per-point repack of a factor's source cloud (12 B point + the six floats of the
symmetric covariance, chunk-major) instead of the API layout's 12 + 36 B -- replaces the per-point reads of
include/gtsam_points/cuda/kernels/vgicp_derivatives.cuh:36-50.  A mirror is only built from covariances that are symmetric to the last bit, so the
six floats ARE the caller's and every record must equal the unmirrored path BIT FOR BIT: that is what these tests hold, on top of the oracle parity the
rest of the suite checks with the mirror on (it is the default)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from helpers import BLOCKS, assert_linearized_close, expmap, rel_err

pytestmark = pytest.mark.gpu
PARITY_TOL = 1e-6
KERNEL, SOURCE_POLICY, FUSED, TILE_CHUNKS, XCD_CHUNK, MIRROR, EFF_MIRROR = 0, 1, 17, 18, 2, 21, 22


def _build(gpu, d, res):
    tgt = gpu.PointCloudGPU(d["target_points"], d["target_covs"])
    src = gpu.PointCloudGPU(d["source_points"], d["source_covs"], normals=d.get("source_normals"))
    vm = gpu.GaussianVoxelMapGPU(res, target_points_drop_rate=0.0)
    vm.insert(tgt)
    return tgt, src, vm


def _lin(gpu, f, delta):
    rec = gpu._capi.Linearized6()
    gpu._capi.check(f._lib.gp_vgicp_factor_linearize(f._h, gpu.types._pose16(delta), C.byref(rec)), "linearize")
    return gpu.LinearizedSystem6(rec)


def _err(gpu, f, dl, de):
    e = C.c_double()
    gpu._capi.check(f._lib.gp_vgicp_factor_compute_error(f._h, gpu.types._pose16(dl), gpu.types._pose16(de), C.byref(e)), "compute_error")
    return e.value


def _batch(gpu, factors, stream=None):
    lib = gpu.load()
    arr = (C.c_void_p * len(factors))(*[f._h.value for f in factors])
    b = C.c_void_p()
    gpu._capi.check(lib.gp_vgicp_batch_create(arr, len(factors), stream, C.byref(b)), "batch")
    return b


def _effective(gpu, batch):
    v = C.c_int(-7)
    gpu._capi.check(gpu.load().gp_vgicp_batch_get_tuning(batch, EFF_MIRROR, C.byref(v)), "get_tuning")
    return v.value


@pytest.mark.parametrize("n_src", [63, 64, 4097, 70_013, 400_077])
@pytest.mark.parametrize("policy", [1, 2])
def test_packed_mirror_is_bit_identical(gpu, n_src, policy):
    """one factor, mirror on (default) vs off: fixed tiles (small), the balanced plan (>= 65536 points), the per-lane tail behind the last full chunk,
    both source-stream cache policies, linearise and error evaluation, fused and two-kernel finalize"""
    from gtsam_points_amd import synthetic

    d = synthetic.make_c2_workload(n_src, 200_000, seed=11)
    _, src, vm = _build(gpu, d, 0.5)
    delta = d["T_true"] @ expmap([2e-4, -1e-4, 1.5e-4, 0.02, -0.01, 0.015])
    de = delta @ expmap([0.002, -0.001, 0.003, 0.01, 0.02, -0.01])
    got = {}
    for mirror in (1, 0):
        f = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(SOURCE_POLICY, policy).set_tuning(MIRROR, mirror)
        L = _lin(gpu, f, delta)
        f.set_tuning(FUSED, 0)
        L2 = _lin(gpu, f, delta)
        for k in BLOCKS:
            assert np.array_equal(getattr(L, k), getattr(L2, k)), (mirror, k)
        got[mirror] = (L, _err(gpu, f, delta, de))
    for k in BLOCKS:
        assert np.array_equal(getattr(got[1][0], k), getattr(got[0][0], k)), k
    assert got[1][0].num_inliers == got[0][0].num_inliers and got[1][0].error == got[0][0].error
    assert got[1][1] == got[0][1]
    om = oracle.OracleVoxelMap(0.5)
    om.insert(d["target_points"], d["target_covs"])
    fo = oracle.OracleVGICPFactor(om, d["source_points"], d["source_covs"], oracle.max_threads())
    assert_linearized_close(got[1][0], fo.linearize(delta), PARITY_TOL, f"mirror, {n_src} points")
    eo = fo.error(de)
    assert abs(got[1][1] - eo) <= PARITY_TOL * abs(eo)


def test_mirror_state_sharing_and_lifetime(gpu, kitti07):
    """GP_TUNE_EFFECTIVE_MIRROR says what a built table streams; factors on one cloud share ONE mirror (gp_source_mirror_bytes grows once) and the
    mirror dies with its last factor; a batch is bit-identical with and without; the k-th factor of a cloud re-reads what the first one left in L2"""
    lib = gpu.load()
    clouds = [gpu.PointCloudGPU(kitti07[f"points_{i}"], kitti07[f"covs_{i}"]) for i in range(5)]
    maps = []
    for c in clouds:
        m = gpu.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        m.insert(c)
        maps.append(m)
    base = lib.gp_source_mirror_bytes()
    pairs = [(t, s) for s in range(5) for t in range(5) if t != s]  # four factors per source cloud
    rng = np.random.default_rng(3)
    poses = np.stack([np.ascontiguousarray(expmap(rng.uniform(-0.05, 0.05, 6)).T).reshape(16) for _ in pairs]).copy()
    out = {}
    for mirror in (1, 0):
        factors = [gpu.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s]) for t, s in pairs]
        b = _batch(gpu, factors)
        assert _effective(gpu, b) == 1  # the default
        if mirror:
            want = sum(((c.size() + 63) // 64) * 2304 for c in clouds)
            assert lib.gp_source_mirror_bytes() - base == want, (lib.gp_source_mirror_bytes(), base, want)
        gpu._capi.check(lib.gp_vgicp_batch_set_tuning(b, MIRROR, mirror), "mirror")
        o = np.zeros((len(pairs), 122))
        gpu._capi.check(lib.gp_vgicp_batch_linearize(b, poses.ctypes.data, o.ctypes.data), "linearize")
        assert _effective(gpu, b) == mirror
        assert lib.gp_vgicp_batch_actual_bytes(b) < lib.gp_vgicp_batch_algorithmic_bytes(b) if mirror else True
        e = np.zeros(len(pairs))
        gpu._capi.check(lib.gp_vgicp_batch_compute_error(b, poses.ctypes.data, poses.ctypes.data, e.ctypes.data), "compute_error")
        out[mirror] = (o, e)
        lib.gp_vgicp_batch_destroy(b)
        del factors
        import gc

        gc.collect()
        assert lib.gp_source_mirror_bytes() == base  # the last factor on a cloud takes the mirror with it
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.allclose(out[1][1], out[1][0][:, 1], rtol=1e-7)  # evaluated at the linearisation pose


def test_unsymmetric_cloud_keeps_the_callers_arrays(gpu, kitti00):
    """a cloud with a covariance that is not symmetric to the last bit gets no mirror (the API arrays carry the (a_ij + a_ji) / 2 symmetrisation in f64):
    the table reports it, and a batch that contains such a factor keeps every factor on the caller's arrays"""
    lib = gpu.load()
    sc = kitti00["source_covs"].copy()
    v = sc[5, 1, 0].view(np.int32) + np.int32(1)
    sc[5, 1, 0] = v.view(np.float32)
    tgt = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])
    good = gpu.PointCloudGPU(kitti00["source_points"], kitti00["source_covs"])
    bad = gpu.PointCloudGPU(kitti00["source_points"], sc)
    vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    base = lib.gp_source_mirror_bytes()
    fg, fb = gpu.IntegratedVGICPFactorGPU(0, 1, vm, good), gpu.IntegratedVGICPFactorGPU(0, 1, vm, bad)
    delta = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    Lg, Lb = _lin(gpu, fg, delta), _lin(gpu, fb, delta)
    assert _effective(gpu, _batch(gpu, [fg])) == 1 and _effective(gpu, _batch(gpu, [fb])) == 0 and _effective(gpu, _batch(gpu, [fg, fb])) == 0
    assert lib.gp_source_mirror_bytes() - base == ((good.size() + 63) // 64) * 2304  # the unusable mirror holds no memory
    om = oracle.OracleVoxelMap(0.5)
    om.insert(kitti00["target_points"], kitti00["target_covs"])
    Lo = oracle.OracleVGICPFactor(om, kitti00["source_points"], sc, 2).linearize(delta)
    for k in ["H_target", "H_source"]:
        h = getattr(Lo, k)
        assert rel_err(getattr(Lb, k), 0.5 * (h + h.T)) <= PARITY_TOL
    assert rel_err(Lb.b_source, Lo.b_source) <= PARITY_TOL
    assert rel_err(Lg.H_source, Lb.H_source) < 1e-6  # (one covariance entry moved by one ulp)


def test_rewritten_source_arrays_repack(gpu, kitti00):
    """the borrowed arrays are immutable while a factor holds them; an owner that rewrites them says so: gp_vgicp_factor_set_source (same pointers) makes the
    factor pack again, gp_source_mirror_invalidate keeps later factors from joining the stale mirror"""
    import torch

    lib = gpu.load()
    tgt = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])
    src = gpu.PointCloudGPU(kitti00["source_points"], kitti00["source_covs"])
    vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    delta = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    f = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src)
    L0 = _lin(gpu, f, delta)
    moved = (kitti00["source_points"] + np.float32(0.05)).astype(np.float32)
    src.points_gpu.copy_(torch.from_numpy(moved).to(src.points_gpu.device))  # in place: same address, new contents
    torch.cuda.synchronize()
    gpu._capi.check(lib.gp_source_mirror_invalidate(C.c_void_p(src.points_gpu.data_ptr())), "invalidate")
    gpu._capi.check(lib.gp_vgicp_factor_set_source(f._h, src.ptr(src.points_gpu), src.ptr(src.covs_gpu), None), "set_source")
    L1 = _lin(gpu, f, delta)
    f2 = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src)  # a later factor on the same addresses
    L2 = _lin(gpu, f2, delta)
    om = oracle.OracleVoxelMap(0.5)
    om.insert(kitti00["target_points"], kitti00["target_covs"])
    Lo = oracle.OracleVGICPFactor(om, moved, kitti00["source_covs"], 2).linearize(delta)
    assert_linearized_close(L1, Lo, PARITY_TOL, "repacked")
    for k in BLOCKS:
        assert np.array_equal(getattr(L1, k), getattr(L2, k))
    assert not np.array_equal(L0.H_source, L1.H_source)


def test_mirror_with_surface_validation_and_offloading(gpu):
    """the normals stay a row of the caller's array next to the packed chunk (K = 4 requests per chunk); offload + reload hands the factor new arrays
    (gp_vgicp_factor_set_source through touch_points) and with them a new mirror: same record"""
    from gtsam_points_amd import synthetic

    lib = gpu.load()
    d = synthetic.make_pair(150_000, 200_000, seed=5)
    d["source_normals"] = d["source_normals"].copy()
    d["source_normals"][::3] *= -1.0
    _, src, vm = _build(gpu, d, 0.5)
    delta = d["T_true"]
    res = {}
    for mirror in (1, 0):
        f = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(MIRROR, mirror)
        f.set_enable_surface_validation(True)
        res[mirror] = _lin(gpu, f, delta)
        if mirror:
            base = lib.gp_source_mirror_bytes()
            assert base >= ((src.size() + 63) // 64) * 2304
            f.set_enable_offloading(True)
            assert src.offload_gpu()
            f.touch_points()  # reloads the cloud (new tensors) and hands the factor the new pointers
            assert src.loaded_on_gpu()
            again = _lin(gpu, f, delta)
            for k in BLOCKS:
                assert np.array_equal(getattr(again, k), getattr(res[1], k)), k
            assert lib.gp_source_mirror_bytes() == base  # the old mirror went with the old arrays, the new one took its place
    for k in BLOCKS:
        assert np.array_equal(getattr(res[1], k), getattr(res[0], k)), k
    assert res[1].num_inliers == res[0].num_inliers and 0 < res[1].num_inliers < 150_000


def test_in_argument_launch_ignores_xcd_chunk(gpu, kitti00):
    """with GP_TUNE_XCD_CHUNK > 0 on a single small factor the grid is rounded up for the chunked map, which the in-argument launch of the stream
    kernel does not use: workgroups beyond an XCD's share ran a neighbour's tile a second time and the fused by-factor finalize counted too many arrivals.
    Now such workgroups leave (and the launch keeps the contiguous map): fused == two-kernel, pose after pose."""
    lib = gpu.load()
    n = 2560  # ten 256-point tiles
    tgt = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])
    src = gpu.PointCloudGPU(kitti00["source_points"][:n], kitti00["source_covs"][:n])
    vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    f = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src)
    b = _batch(gpu, [f])
    gpu._capi.check(lib.gp_vgicp_batch_set_tuning(b, XCD_CHUNK, 4), "xcd chunk")
    rng = np.random.default_rng(1)
    out = np.zeros((1, 122))
    for it in range(6):
        pose = np.ascontiguousarray(expmap(rng.uniform(-0.02, 0.02, 6)).T).reshape(1, 16).copy()
        recs = []
        for fused in (1, 0, 1):
            gpu._capi.check(lib.gp_vgicp_batch_set_tuning(b, FUSED, fused), "fused")
            gpu._capi.check(lib.gp_vgicp_batch_linearize(b, pose.ctypes.data, out.ctypes.data), "linearize")
            recs.append(out.copy())
        assert np.array_equal(recs[0], recs[1]) and np.array_equal(recs[0], recs[2]), it
    lib.gp_vgicp_batch_destroy(b)


def test_reestimated_covariances_do_not_join_a_stale_mirror(gpu, kitti00):
    """estimate_covariances_gpu replaces frame.covs_gpu; torch's caching allocator hands the freed block out again at the same address, so a factor
    created afterwards could join the packed mirror of the OLD covariances. The frame forgets its mirrors and bumps its generation: old and new factor both
    linearise with the new covariances; PointCloudGPU.contents_changed() does the same for tensors rewritten in place."""
    import torch

    from gtsam_points_amd.features import estimate_covariances_gpu

    tgt = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])
    vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    delta = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    src = gpu.PointCloudGPU(kitti00["source_points"], kitti00["source_covs"])
    f_old = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src)
    L_old = _lin(gpu, f_old, delta)  # builds the mirror of (points, k = 10 covariances)
    gen = src.generation
    for k in (5, 20):  # same-size tensors: the allocator recycles the block that was just freed
        estimate_covariances_gpu(src, k_neighbors=k)
    torch.cuda.synchronize()
    assert src.generation > gen
    covs_new = src.covs_gpu.cpu().numpy()
    f_new = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src)
    L_new = _lin(gpu, f_new, delta)
    f_plain = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src).set_tuning(MIRROR, 0)
    L_plain = _lin(gpu, f_plain, delta)
    for k in BLOCKS:
        assert np.array_equal(getattr(L_new, k), getattr(L_plain, k)), k
    assert not np.array_equal(L_new.H_source, L_old.H_source)
    om = oracle.OracleVoxelMap(0.5)
    om.insert(kitti00["target_points"], kitti00["target_covs"])
    Lo = oracle.OracleVGICPFactor(om, kitti00["source_points"], covs_new, 2).linearize(delta)
    assert_linearized_close(L_new, Lo, PARITY_TOL, "re-estimated covariances")
    # in place, through the wrapper's own hook
    src.covs_gpu.copy_(torch.from_numpy(np.ascontiguousarray(kitti00["source_covs"].reshape(-1, 9))).to(src.covs_gpu.device))
    torch.cuda.synchronize()
    src.contents_changed()
    L_back = _lin(gpu, gpu.IntegratedVGICPFactorGPU(0, 1, vm, src), delta)
    for k in BLOCKS:
        assert np.array_equal(getattr(L_back, k), getattr(L_old, k)), k
