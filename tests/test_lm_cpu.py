"""bench_lm.py (the harness behind bench.py's configs.lm_*): the reference's LM cadence (src/gtsam_points/optimizers/levenberg_marquardt_ext.cpp:188-392) restated over the
CHECKER's CPU factors reaches the reference's alignment gate (src/test/test_matching_cost_factors.cpp:227: < 0.015 rad / 0.15 m) on the kitti_07_dump graph -- the
CPU leg of the bench object, and the proof that the harness itself optimises correctly before it is pointed at the GPU path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench_lm  # noqa: E402
import oracle  # noqa: E402


def test_expmap_and_inverse_match_the_scalar_forms():
    from gtsam_points_amd.synthetic import expmap

    rng = np.random.default_rng(1)
    xi = np.concatenate([rng.uniform(-1.5, 1.5, (50, 6)), np.zeros((1, 6)), 1e-10 * rng.normal(size=(3, 6))])
    T = bench_lm.expmap_many(xi)
    for k in range(len(xi)):
        assert np.abs(T[k] - expmap(xi[k])).max() < 1e-12
    assert np.abs(bench_lm.inv_many(T) @ T - np.eye(4)).max() < 1e-12


def test_cpu_lm_reaches_the_alignment_gate(kitti07):
    n = 5
    maps = []
    for i in range(n):
        m = oracle.OracleVoxelMap(1.0)
        m.insert(kitti07[f"points_{i}"], kitti07[f"covs_{i}"])
        maps.append(m)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]  # the demo's ten factors (demo_benchmark.cpp:171-173)
    factors = [oracle.OracleVGICPFactor(maps[i], kitti07[f"points_{j}"], kitti07[f"covs_{j}"], 4) for i, j in pairs]
    truth = np.stack([np.asarray(T, dtype=np.float64) for T in kitti07["poses"][:n]])
    rng = np.random.default_rng(8191)
    v0 = truth @ bench_lm.expmap_many(rng.uniform(-0.1, 0.1, (n, 6)))  # test_matching_cost_factors.cpp:86-92
    v0[0] = truth[0]
    g = bench_lm.CpuGraph(factors, pairs, n, fixed=0)
    res = bench_lm.run_lm(g, v0, max_iterations=30)
    s = bench_lm.summarize(res, g, truth, "cpu")
    assert s["gate_met"], s
    assert all(after < before for before, after in res["steps"])  # every accepted step lowered the cost it was measured against (a re-linearisation may raise it again:
    # the correspondences change -- levenberg_marquardt_ext.cpp:410, "error can increase due to data association changes")
    assert 2 <= s["iterations"] <= 30 and s["inner_iterations"] >= s["iterations"]
    assert set(s["ms_per_iteration_by_phase"]) == {"linearize", "solve", "error", "glue"}
    # the host system builder of the harness against a direct sum over the records
    A, b, c = bench_lm.host_system(g.rec, g.factor_slots, g.num_slots)
    assert np.abs(A - A.T).max() < 1e-9 * np.abs(A).max() and abs(c - g.rec[:, 1].sum()) < 1e-9 * abs(c)
    assert np.abs(b[:6] + sum(g.rec[k, 110:116] * (g.factor_slots[k, 0] == 0) + g.rec[k, 116:122] * (g.factor_slots[k, 1] == 0) for k in range(len(pairs)))).max() < 1e-9 * np.abs(b[:6]).max()
