"""Pins the C oracle against the REFERENCE's own code: oracle/_ref/libref.so is built from /root/reference's
integrated_vgicp_factor_impl.hpp, scan_matching_reduction.hpp, integrated_matching_cost_factor.cpp, gaussian_voxelmap_cpu.cpp,
incremental_voxelmap_impl.hpp and fast_floor.hpp (where they lie; only Eigen/GTSAM are stand-ins, oracle/ref_shim/include)."""
import numpy as np
import pytest

import oracle
from oracle import refcapi
from helpers import BLOCKS, assert_linearized_close, expmap, rel_err

pytestmark = pytest.mark.skipif(not refcapi.available(), reason="oracle/_ref/libref.so not built (needs /root/reference at build time)")


def _both(d, res, threads=1):
    om = oracle.OracleVoxelMap(res)
    om.insert(d["target_points"], d["target_covs"])
    rm = refcapi.RefVoxelMap(res)
    rm.insert(d["target_points"], d["target_covs"])
    return (om, oracle.OracleVGICPFactor(om, d["source_points"], d["source_covs"], threads)), (rm, refcapi.RefVGICPFactor(rm, d["source_points"], d["source_covs"], threads))


@pytest.mark.parametrize("res", [0.5, 1.0, 0.3])
def test_oracle_equals_reference_code_on_kitti00(kitti00, res):
    (om, fo), (rm, fr) = _both(kitti00, res)
    assert om.num_voxels == rm.num_voxels
    for xi in [np.zeros(6), [0.01, -0.02, 0.015, 0.10, -0.05, 0.03], [0.05, 0.04, -0.06, -0.5, 0.3, 0.2]]:
        delta = expmap(xi)
        Lo, Lr = fo.linearize(delta), fr.linearize(delta)
        assert_linearized_close(Lo, Lr, 1e-12, f"res {res} xi {xi}")
        # error() with correspondences / Mahalanobis frozen at the linearisation point
        de = delta @ expmap([0.002, -0.001, 0.003, 0.01, 0.02, -0.01])
        eo, er = fo.error(de), fr.error(de)
        assert abs(eo - er) <= 1e-12 * abs(er)


def test_golden_vectors_match_reference_code(kitti00, kitti07, golden):
    """the committed golden vectors (generated from the C oracle) equal what the reference's own classes compute"""
    for name, res in [("kitti00_dec8_r0.5_identity", 0.5), ("kitti00_dec8_r0.5_c1b", 0.5), ("kitti00_dec8_r1.0_c1b", 1.0)]:
        g = golden[name]
        _, (rm, fr) = _both(kitti00, res)
        assert rm.num_voxels == g["num_voxels"]
        assert_linearized_close(fr.linearize(np.array(g["delta"])), g, 1e-12, name)
    for i in range(4):
        g = golden[f"kitti07_dec4_{i}_{i+1}_r1.0"]  # poses from 6-digit quaternions: R not orthonormal, used as given
        d = dict(target_points=kitti07[f"points_{i}"], target_covs=kitti07[f"covs_{i}"], source_points=kitti07[f"points_{i+1}"], source_covs=kitti07[f"covs_{i+1}"])
        _, (_, fr) = _both(d, 1.0)
        assert_linearized_close(fr.linearize(np.array(g["delta"])), g, 1e-12, g["name"])


def test_reference_code_threads_and_synthetic(kitti00):
    from gtsam_points_amd import synthetic

    d = synthetic.make_pair(30000, 60000, seed=11)
    (om, fo), (rm, fr) = _both(d, 0.5, threads=4)
    assert om.num_voxels == rm.num_voxels
    delta = d["T_true"] @ expmap([0.002, -0.001, 0.0015, 0.02, -0.01, 0.015])
    assert_linearized_close(fo.linearize(delta), fr.linearize(delta), 1e-11, "synthetic, 4 threads")
