"""the optimizer iteration, not just the call.  bench_lm.py's LM loop (the reference's cadence, levenberg_marquardt_ext.cpp:188-392) over the GPU path --
ONE batched linearise per iteration, records resident in HBM, block-sparse LL^T on the device, ONE batched error evaluation per trial -- reaches the
reference's alignment gate (test_matching_cost_factors.cpp:227) on the kitti_07_dump graph, ends where the same loop over the checker's CPU factors ends, and the
host-solve variant agrees with the device solve."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench_lm  # noqa: E402
import oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def test_gpu_lm_reaches_the_gate_and_the_cpu_result(gpu, kitti07):
    n = 5
    clouds = [gpu.PointCloudGPU(kitti07[f"points_{i}"], kitti07[f"covs_{i}"]) for i in range(n)]
    maps, omaps = [], []
    for i, c in enumerate(clouds):
        vm = gpu.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        vm.insert(c)
        maps.append(vm)
        om = oracle.OracleVoxelMap(1.0)
        om.insert(kitti07[f"points_{i}"], kitti07[f"covs_{i}"])
        omaps.append(om)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    factors = [gpu.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j]) for i, j in pairs]
    cpu_factors = [oracle.OracleVGICPFactor(omaps[i], kitti07[f"points_{j}"], kitti07[f"covs_{j}"], 4) for i, j in pairs]
    truth = np.stack([np.asarray(T, dtype=np.float64) for T in kitti07["poses"][:n]])
    v0 = truth @ bench_lm.expmap_many(np.random.default_rng(8191).uniform(-0.1, 0.1, (n, 6)))
    v0[0] = truth[0]
    cg = bench_lm.CpuGraph(cpu_factors, pairs, n, fixed=0)
    res_cpu = bench_lm.run_lm(cg, v0, max_iterations=30)
    assert bench_lm.summarize(res_cpu, cg, truth, "cpu")["gate_met"]
    results = {}
    for solver in ("device", "device-three-calls", "host"):
        gg = bench_lm.GpuGraph(gpu, factors, pairs, n, fixed=0, solver=solver)
        res = bench_lm.run_lm(gg, v0, max_iterations=30)
        s = bench_lm.summarize(res, gg, truth, solver)
        assert s["gate_met"], s
        assert all(after < before for before, after in res["steps"])
        assert s["iterations"] == bench_lm.summarize(res_cpu, cg, truth, "cpu")["iterations"], (s, res_cpu["errors"])  # the same decisions at every step
        for k in range(n):
            ang, tr = bench_lm.pose_error(res["values"][k], res_cpu["values"][k])
            assert ang < 1e-5 and tr < 1e-4, (solver, k, ang, tr)
        assert abs(res["final_error"] - res_cpu["final_error"]) < 1e-5 * res_cpu["final_error"]
        gg.sync_phases = True
        again = bench_lm.run_lm(gg, v0, max_iterations=30)
        assert again["errors"] == res["errors"]  # bit-reproducible, with or without the wait between linearise and solve
        results[solver] = res
        gg.close()
    assert results["device"]["errors"] == results["device-three-calls"]["errors"] and np.array_equal(results["device"]["values"], results["device-three-calls"]["values"])  # one call = the three
    for k in range(n):
        ang, tr = bench_lm.pose_error(results["device"]["values"][k], results["host"]["values"][k])
        assert ang < 1e-7 and tr < 1e-6


def _kitti_graph(gpu, kitti07, n=5, pairs=None):
    clouds = [gpu.PointCloudGPU(kitti07[f"points_{i}"], kitti07[f"covs_{i}"]) for i in range(n)]
    maps = []
    for c in clouds:
        vm = gpu.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        vm.insert(c)
        maps.append(vm)
    pairs = pairs or [(i, j) for i in range(n) for j in range(i + 1, n)]
    factors = [gpu.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j]) for i, j in pairs]
    truth = np.stack([np.asarray(T, dtype=np.float64) for T in kitti07["poses"][:n]])
    v0 = truth @ bench_lm.expmap_many(np.random.default_rng(8191).uniform(-0.1, 0.1, (n, 6)))
    v0[0] = truth[0]
    return factors, pairs, truth, v0, (clouds, maps)


def _rigid(values):
    """nearest rotations (the fixture's poses come from 6-digit text: orthonormal to 1e-6, which sends the host-pose entry points to the general kernels)"""
    out = np.array(values, dtype=np.float64)
    for T in out:
        u, _, vt = np.linalg.svd(T[:3, :3])
        T[:3, :3] = u @ vt
    return out


@pytest.mark.parametrize("single,rigid", [(False, True), (True, True), (False, False), (True, False)])
def test_device_pose_tables_give_the_same_records(gpu, kitti07, single, rigid):
    """gp_vgicp_batch_issue_linearize_dev / _compute_error_dev (poses already in HBM) == the host-pose entry points, bit for bit -- a batch, and a batch of ONE (whose
    host-pose form carries pose and descriptor in the kernel arguments: another instantiation)"""
    import ctypes as C

    import torch
    from gtsam_points_amd import _capi

    factors, pairs, truth, v0, keep = _kitti_graph(gpu, kitti07)
    if single:
        factors, pairs = factors[:1], pairs[:1]
    lib = gpu.load()
    F = len(factors)
    g = bench_lm._Graph(pairs, 5)
    if rigid:
        truth, v0 = _rigid(truth), _rigid(v0)
    p_lin, p_eval = bench_lm._poses16(g.deltas(v0)), bench_lm._poses16(g.deltas(truth))
    batch = C.c_void_p()
    _capi.check(lib.gp_vgicp_batch_create((C.c_void_p * F)(*[f._h.value for f in factors]), F, None, C.byref(batch)), "batch")
    rec = [torch.zeros((F, 122), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    err = [torch.zeros(F, dtype=torch.float64, device="cuda:0") for _ in range(2)]
    d_lin, d_eval = torch.from_numpy(p_lin).cuda(), torch.from_numpy(p_eval).cuda()
    torch.cuda.synchronize()
    _capi.check(lib.gp_vgicp_batch_issue_linearize(batch, p_lin.ctypes.data, C.c_void_p(rec[0].data_ptr())), "host poses")
    _capi.check(lib.gp_vgicp_batch_issue_linearize_dev(batch, C.c_void_p(d_lin.data_ptr()), int(rigid), C.c_void_p(rec[1].data_ptr())), "device poses")
    _capi.check(lib.gp_vgicp_batch_issue_compute_error(batch, p_lin.ctypes.data, p_eval.ctypes.data, C.c_void_p(err[0].data_ptr())), "host poses")
    _capi.check(lib.gp_vgicp_batch_issue_compute_error_dev(batch, C.c_void_p(d_lin.data_ptr()), C.c_void_p(d_eval.data_ptr()), C.c_void_p(err[1].data_ptr())), "device poses")
    _capi.check(lib.gp_vgicp_batch_sync(batch), "sync")
    st = C.c_void_p(1)
    _capi.check(lib.gp_vgicp_batch_stream(batch, C.byref(st)), "stream")
    assert not st.value  # created on the null stream
    lib.gp_vgicp_batch_destroy(batch)
    assert rec[0][:, 0].min().item() > 100  # inliers: something was matched
    assert torch.equal(rec[0], rec[1]) and torch.equal(err[0], err[1])


@pytest.mark.parametrize("rigid", [True, False])
def test_trial_on_the_device_follows_the_host_driven_loop(gpu, kitti07, rigid):
    """gp_lm_graph_* (values in HBM: damped step + retract + error evaluation, one wait per trial) against the host-driven loop of the test above: the same decisions, the same
    costs to rounding, the same result; the library's own loop (gp_lm_graph_optimize) == the interpreter driving its three calls, bit for bit."""
    factors, pairs, truth, v0, keep = _kitti_graph(gpu, kitti07)
    if rigid:
        truth, v0 = _rigid(truth), _rigid(v0)
    gg = bench_lm.GpuGraph(gpu, factors, pairs, 5, fixed=0, solver="device")
    ref = bench_lm.run_lm(gg, v0, max_iterations=30)
    tg = bench_lm.GpuTrialGraph(gpu, factors, pairs, 5, fixed=0)
    res = bench_lm.run_lm(tg, v0, max_iterations=30)
    s = bench_lm.summarize(res, tg, truth, "trial")
    assert s["gate_met"], s
    assert res["iterations"] == ref["iterations"] and res["inner_iterations"] == ref["inner_iterations"]
    np.testing.assert_allclose(res["errors"], ref["errors"], rtol=1e-9)
    np.testing.assert_allclose(res["values"], ref["values"], atol=1e-9)
    # one trial, piece by piece: the step of the records at v0, numpy's retract, the batch's error evaluation at those values
    tg.g.set_values(v0)
    tg.g.linearize()
    dx, b, c, e, vt = tg.g.try_lambda(1e-3, want_values=True)
    gg.linearize(v0)
    dx0, b0, c0 = gg.solve(1e-3)
    np.testing.assert_allclose(dx, dx0, rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(b, b0, rtol=1e-10, atol=1e-12)
    assert abs(c - c0) <= 1e-12 * c0
    np.testing.assert_allclose(vt, gg.retract(v0, dx0), atol=1e-12)
    assert abs(e - gg.error(gg.retract(v0, dx0))) <= 1e-9 * e
    assert np.array_equal(vt[0], v0[0])  # the held pose
    # the library's loop
    nat = tg.native_loop(v0, max_iterations=30)
    assert nat["iterations"] == res["iterations"] and nat["inner_iterations"] == res["inner_iterations"]
    assert nat["final_error"] == res["final_error"] and np.array_equal(nat["values"], res["values"])
    again = tg.native_loop(v0, max_iterations=30)
    assert np.array_equal(again["values"], nat["values"])  # bit-reproducible
    # speculation (the linearise at the trial values queued behind every trial) changes no bit: the loop without it, and a REJECTED trial -- two trials in a row from
    # the same linearisation, then the accepted one's next linearisation -- with and without
    assert tg.g.set_speculation(False) is True
    plain = tg.native_loop(v0, max_iterations=30)
    assert plain["final_error"] == nat["final_error"] and np.array_equal(plain["values"], nat["values"])
    seq = {}
    for spec in (True, False):
        tg.g.set_speculation(spec)
        tg.g.set_values(v0)
        tg.g.linearize()
        a = [np.array(x, copy=True) for x in tg.g.try_lambda(1e-5)]
        b2 = [np.array(x, copy=True) for x in tg.g.try_lambda(1e-1)]  # (the first one is dropped: as a rejected step would be)
        tg.g.accept()
        tg.g.linearize()
        c2 = [np.array(x, copy=True) for x in tg.g.try_lambda(1e-3)]
        seq[spec] = a + b2 + c2 + [tg.g.values()]
    assert all(np.array_equal(x, y) for x, y in zip(seq[True], seq[False]))
    assert not np.array_equal(seq[True][0], seq[True][4])  # (the two lambdas gave different steps)
    # the multi-launch step with the retract as a kernel of its own behind it (what a graph too large for the one-launch step runs): the same bits
    assert tg.g.set_one_launch(False) == 0
    multi = tg.native_loop(v0, max_iterations=30)
    assert multi["final_error"] == nat["final_error"] and np.array_equal(multi["values"], nat["values"])
    assert tg.g.set_one_launch(True) == 1
    gg.close()
    tg.close()


def test_trial_graph_with_one_free_pose_and_without_a_gauge(gpu, kitti07):
    """the dense 6 x 6 step behind the same calls (two poses, one held); a graph nobody holds reports the indeterminate system at lambda 0 -- b and c stay valid -- and takes
    the next, damped trial"""
    factors, pairs, truth, v0, keep = _kitti_graph(gpu, kitti07, n=2, pairs=[(0, 1)])
    gg = bench_lm.GpuGraph(gpu, factors, pairs, 2, fixed=0, solver="device")
    ref = bench_lm.run_lm(gg, v0, max_iterations=30)
    tg = bench_lm.GpuTrialGraph(gpu, factors, pairs, 2, fixed=0)
    res = bench_lm.run_lm(tg, v0, max_iterations=30)
    assert res["iterations"] == ref["iterations"] and bench_lm.summarize(res, tg, truth, "trial")["gate_met"]
    np.testing.assert_allclose(res["values"], ref["values"], atol=1e-9)
    nat = tg.native_loop(v0, max_iterations=30)
    assert np.array_equal(nat["values"], res["values"])
    assert tg.g.set_one_launch(False) == 0  # (the 6 x 6 system's nine-operation path, lm_poses_kernel behind it)
    assert np.array_equal(tg.native_loop(v0, max_iterations=30)["values"], nat["values"])
    gg.close()
    tg.close()
    free = gpu.LevenbergMarquardtGraphGPU(factors, pairs, 2, fixed=())
    assert free.n == 12
    with pytest.raises(gpu.GPError, match="linearize first"):
        free.try_lambda(1.0)
    free.set_values(v0)
    free.linearize()
    with pytest.raises(gpu.GPError, match="indeterminate"):
        free.try_lambda(0.0)
    assert free._c[0] > 0 and np.abs(free._b).max() > 0
    with pytest.raises(gpu.GPError, match="no successful trial"):
        free.accept()
    dx, b, c, e = free.try_lambda(1e-3)
    assert np.isfinite(dx).all() and e < c
    free.accept()
    assert np.abs(free.values() - v0).max() > 1e-4
    with pytest.raises(gpu.GPError, match="lambda I damping only"):
        free.optimize(diagonal_damping=1)
    free.close()


def test_one_large_factor_through_the_device_pose_entry_points(gpu):
    """a single factor of >= 65536 points is a PLANNED batch (its tile list is the balanced stream plan, its host-pose calls carry pose and descriptor in the kernel
    arguments): the device-pose entry points read the same plan out of the tile table -- same records, same error, bit for bit -- and the LM graph over it converges"""
    import ctypes as C

    import torch
    from gtsam_points_amd import _capi, synthetic

    d = synthetic.make_pair(120000, 200000, seed=5)
    tgt, src = gpu.PointCloudGPU(d["target_points"], d["target_covs"]), gpu.PointCloudGPU(d["source_points"], d["source_covs"])
    vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    f = gpu.IntegratedVGICPFactorGPU(0, 1, vm, src)
    lib = gpu.load()
    delta = d["T_true"] @ synthetic.expmap([0.004, -0.002, 0.003, 0.03, -0.02, 0.025])
    p_lin, p_eval = bench_lm._poses16(delta[None]), bench_lm._poses16(d["T_true"][None])
    batch = C.c_void_p()
    _capi.check(lib.gp_vgicp_batch_create((C.c_void_p * 1)(f._h.value), 1, None, C.byref(batch)), "batch")
    rec = [torch.zeros((1, 122), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    err = [torch.zeros(1, dtype=torch.float64, device="cuda:0") for _ in range(2)]
    e_sync = np.zeros(1)
    d_lin, d_eval = torch.from_numpy(p_lin).cuda(), torch.from_numpy(p_eval).cuda()
    torch.cuda.synchronize()
    _capi.check(lib.gp_vgicp_batch_issue_linearize(batch, p_lin.ctypes.data, C.c_void_p(rec[0].data_ptr())), "host pose")
    _capi.check(lib.gp_vgicp_batch_issue_linearize_dev(batch, C.c_void_p(d_lin.data_ptr()), 1, C.c_void_p(rec[1].data_ptr())), "device pose")
    _capi.check(lib.gp_vgicp_batch_issue_compute_error(batch, p_lin.ctypes.data, p_eval.ctypes.data, C.c_void_p(err[0].data_ptr())), "host pose")
    _capi.check(lib.gp_vgicp_batch_issue_compute_error_dev(batch, C.c_void_p(d_lin.data_ptr()), C.c_void_p(d_eval.data_ptr()), C.c_void_p(err[1].data_ptr())), "device pose")
    _capi.check(lib.gp_vgicp_batch_compute_error_dev(batch, C.c_void_p(d_lin.data_ptr()), C.c_void_p(d_eval.data_ptr()), e_sync.ctypes.data), "device pose, polled")
    _capi.check(lib.gp_vgicp_batch_sync(batch), "sync")
    lib.gp_vgicp_batch_destroy(batch)
    assert rec[0][0, 0].item() > 100000 and torch.equal(rec[0], rec[1]) and torch.equal(err[0], err[1]) and e_sync[0] == err[0].item()
    lm = gpu.LevenbergMarquardtGraphGPU([f], [(0, 1)], 2, fixed=(0,))
    values, s = lm.optimize(np.stack([np.eye(4), delta]))
    ang, tr = bench_lm.pose_error(values[1], d["T_true"])
    assert s["iterations"] >= 2 and not s["gave_up"] and ang < 2e-3 and tr < 2e-2, (s, ang, tr)
    lm.close()


def test_rejected_trials_take_the_same_path_in_both_loops(gpu, kitti07):
    """model-fidelity bars the steps miss -- 1.99: some trials are dropped and lambda climbs before a step is taken; 3: every trial is dropped until lambda reaches its
    upper bound and the loop gives up where it started -- each dropped trial with the speculative linearise behind it thrown away: the library's loop and the interpreter
    driving its three calls take the same decisions and end on the same bits, with and without speculation"""
    factors, pairs, truth, v0, keep = _kitti_graph(gpu, kitti07)
    truth, v0 = _rigid(truth), _rigid(v0)
    tg = bench_lm.GpuTrialGraph(gpu, factors, pairs, 5, fixed=0)
    rejected = 0
    for bar in (1.99, 3.0):  # (the factors' cost is r^T M r without the 1/2 of the quadratic model: a perfect step has fidelity 2)
        res = bench_lm.run_lm(tg, v0, max_iterations=12, min_fidelity=bar)
        rejected += res["inner_iterations"] - res["iterations"]
        out = {}
        for spec in (True, False):
            tg.g.set_speculation(spec)
            tg.g.set_values(v0)
            values, s = tg.g.optimize(max_iterations=12, min_model_fidelity=bar)
            out[spec] = values
            assert s["iterations"] == res["iterations"] and s["inner_iterations"] == res["inner_iterations"], (bar, s, res["iterations"], res["inner_iterations"])
            assert s["final_lambda"] == res["final_lambda"] and np.array_equal(values, res["values"]), (bar, s)
        assert np.array_equal(out[True], out[False])
        if bar > 2.0:
            # (no step is ever taken: the search ends when lambda reaches its bound or -- tryLambda's other exit -- when a damped step no longer changes the cost)
            assert res["iterations"] == 1 and res["inner_iterations"] >= 5 and np.array_equal(values, v0), (s, res["inner_iterations"])
    assert rejected >= 5
    tg.close()
