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
