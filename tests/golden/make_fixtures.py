"""Generates the committed fixtures under tests/golden/ (run once in the build container, where /root/reference exists).

  python tests/golden/make_fixtures.py

Inputs : decimated subsets of the reference's bundled scans (data/kitti_00, data/kitti_07_dump; CC BY-NC-SA, see
         data/IMPORTANT_NOTES in the reference -- test data, non-commercial use) + covariances from the oracle's
         estimate_covariances (k=10), stored as float32 exactly as the GPU API receives them.
Outputs: golden linearisations computed by the C oracle (1 thread) and cross-checked here against the independent
         numpy restatement (<= 1e-10 relative) before being written.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402
from oracle import vgicp_oracle_np as onp  # noqa: E402

REF = "/root/reference/data"
C1B = [0.01, -0.02, 0.015, 0.10, -0.05, 0.03]


def sym32(c):
    c = c.astype(np.float32)
    return np.ascontiguousarray(0.5 * (c + c.transpose(0, 2, 1)))


def covs_of(points):
    c, short = oracle.estimate_covariances(points, 10, 8)
    assert short == 0
    return sym32(c)


def quat_to_T(v):
    tx, ty, tz, qx, qy, qz, qw = v
    R = np.array(
        [
            [1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
            [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
            [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)],
        ]
    )
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = [tx, ty, tz]
    return T


def golden_case(name, res, tp, tc, sp, sc, delta, delta_eval=None):
    vm = oracle.OracleVoxelMap(res)
    vm.insert(tp, tc)
    f = oracle.OracleVGICPFactor(vm, sp, sc, 1)
    L = f.linearize(delta)
    vn = onp.VoxelMapNP(res)
    vn.insert(tp, tc.transpose(0, 2, 1).reshape(-1, 9))
    Ln = onp.vgicp_linearize(vn, sp, sc.transpose(0, 2, 1).reshape(-1, 9), delta)
    for k in ["H_target", "H_source", "H_target_source", "b_target", "b_source"]:
        a, b = getattr(L, k), Ln[k]
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-300)
        assert rel < 1e-10, (name, k, rel)
    assert L.num_inliers == Ln["num_inliers"] and vm.num_voxels == vn.num_voxels
    out = dict(
        name=name,
        resolution=res,
        delta=np.asarray(delta).tolist(),
        num_voxels=vm.num_voxels,
        num_inliers=L.num_inliers,
        error=L.error,
        H_target=L.H_target.tolist(),
        H_source=L.H_source.tolist(),
        H_target_source=L.H_target_source.tolist(),
        b_target=L.b_target.tolist(),
        b_source=L.b_source.tolist(),
    )
    if delta_eval is not None:
        out["delta_eval"] = np.asarray(delta_eval).tolist()
        out["error_eval"] = f.error(delta_eval)
        Le = onp.vgicp_linearize(vn, sp, sc.transpose(0, 2, 1).reshape(-1, 9), delta, delta_eval)
        assert abs(out["error_eval"] - Le["error"]) <= 1e-10 * abs(Le["error"])
    print(name, "voxels", vm.num_voxels, "inliers", L.num_inliers, "error", L.error)
    return out


def main():
    cases = []
    # ---- kitti_00, every 8th point ----
    t_full = np.fromfile(f"{REF}/kitti_00/000000.bin", dtype=np.float32).reshape(-1, 3)
    s_full = np.fromfile(f"{REF}/kitti_00/000001.bin", dtype=np.float32).reshape(-1, 3)
    tp, sp = np.ascontiguousarray(t_full[::8]), np.ascontiguousarray(s_full[::8])
    tc, sc = covs_of(tp), covs_of(sp)
    np.savez_compressed(os.path.join(HERE, "kitti00_dec8.npz"), target_points=tp, target_covs=tc, source_points=sp, source_covs=sc)
    d1 = oracle.expmap(C1B)
    d2 = oracle.expmap([0.012, -0.018, 0.013, 0.12, -0.04, 0.02])
    cases.append(golden_case("kitti00_dec8_r0.5_identity", 0.5, tp, tc, sp, sc, np.eye(4)))
    cases.append(golden_case("kitti00_dec8_r0.5_c1b", 0.5, tp, tc, sp, sc, d1, d2))
    cases.append(golden_case("kitti00_dec8_r1.0_c1b", 1.0, tp, tc, sp, sc, d1, d2))
    # ---- full kitti_00 (inputs stay in /root/reference; outputs pin the oracle on the C1 anchor) ----
    tcf, scf = covs_of(t_full), covs_of(s_full)
    cases.append(golden_case("kitti00_full_r0.5_identity", 0.5, t_full, tcf, s_full, scf, np.eye(4)))
    cases.append(golden_case("kitti00_full_r0.5_c1b", 0.5, t_full, tcf, s_full, scf, d1, d2))
    # ---- kitti_07_dump, every 4th point, 5 submaps ----
    poses = [quat_to_T([float(x) for x in line.split()[1:]]) for line in open(f"{REF}/kitti_07_dump/graph.txt")]
    sub = {}
    for i in range(5):
        p = np.fromfile(f"{REF}/kitti_07_dump/{i:06d}/points.bin", dtype=np.float32).reshape(-1, 3)
        p = np.ascontiguousarray(p[::4])
        sub[f"points_{i}"] = p
        sub[f"covs_{i}"] = covs_of(p)
    sub["poses"] = np.stack(poses)
    np.savez_compressed(os.path.join(HERE, "kitti07_dec4.npz"), **sub)
    rng = np.random.default_rng(8191)
    for i in range(4):
        noise = oracle.expmap(rng.uniform(-0.05, 0.05, 6))
        delta = oracle.calc_delta(poses[i], poses[i + 1] @ noise)
        cases.append(golden_case(f"kitti07_dec4_{i}_{i+1}_r1.0", 1.0, sub[f"points_{i}"], sub[f"covs_{i}"], sub[f"points_{i+1}"], sub[f"covs_{i+1}"], delta))
    with open(os.path.join(HERE, "golden_vgicp.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_fixtures.py", oracle="oracle/vgicp_oracle.c (1 thread)", cases=cases), f, indent=1)


if __name__ == "__main__":
    main()
