"""GPU tests of the callers either side of the VGICP path: the overlap_gpu overload set,
merge_frames_gpu and the PointCloudGPU upload path -- modelled on src/test/test_voxelmap.cpp:155-165,211-259."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle
from helpers import expmap

pytestmark = pytest.mark.gpu


def _oracle_hits(om, points, covs, T):
    """per-point "falls in a voxel" mask from the CPU factor's correspondence search (integrated_vgicp_factor_impl.hpp:99-172)"""
    f = oracle.OracleVGICPFactor(om, points, covs, 1)
    f.update_correspondences(T)
    return np.asarray(f.correspondences()) >= 0


def _map(gpu, points, covs, res):
    vm = gpu.GaussianVoxelMapGPU(res, target_points_drop_rate=0.0)
    vm.insert(gpu.PointCloudGPU(points, covs))
    om = oracle.OracleVoxelMap(res)
    om.insert(points, covs)
    return vm, om


def test_overlap_union_and_batch_match_cpu(gpu, kitti00, kitti07):
    """overlap_gpu(targets, source, deltas) counts a point once if ANY target holds it (bool_or_kernel,
    gaussian_voxelmap_gpu_funcs.cu:180-182,265-335); the (targets, sources, deltas) form returns one rate per pair (:337-404)"""
    clouds = [(kitti00["target_points"], kitti00["target_covs"]), (kitti00["source_points"], kitti00["source_covs"]),
              (kitti07["points_0"], kitti07["covs_0"])]
    maps = [_map(gpu, p, c, r) for (p, c), r in zip(clouds, [0.5, 1.0, 0.5])]
    deltas = [expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03]), expmap([0.2, -0.1, 0.3, 1.0, -2.0, 0.5]), np.eye(4)]
    sp, sc = kitti00["source_points"], kitti00["source_covs"]
    src = gpu.PointCloudGPU(sp, sc)
    masks = [_oracle_hits(om, sp, sc, T) for (_, om), T in zip(maps, deltas)]
    union = gpu.overlap_gpu([vm for vm, _ in maps], src, deltas)
    assert round(union * len(sp)) == int(np.logical_or.reduce(masks).sum())
    # a union over one target is the single-target overload
    assert gpu.overlap_gpu([maps[0][0]], src, [deltas[0]]) == gpu.overlap_gpu(maps[0][0], src, deltas[0])
    # pairwise batch: different sources, one launch
    sources = [src, gpu.PointCloudGPU(*clouds[0]), gpu.PointCloudGPU(*clouds[2])]
    host = [(sp, sc), clouds[0], clouds[2]]
    rates = gpu.overlap_gpu([vm for vm, _ in maps], sources, deltas)
    assert len(rates) == 3
    for (vm, om), s, (hp, hc), T, r in zip(maps, sources, host, deltas, rates):
        assert round(r * len(hp)) == int(_oracle_hits(om, hp, hc, T).sum())
        assert r == gpu.overlap_gpu(vm, s, T)
    assert gpu.overlap_gpu([], [], []) == []
    with pytest.raises(gpu.GPError):
        gpu.overlap_gpu([maps[0][0]], [src, src], deltas[:1])  # size mismatch aborts upstream (:342-345)


@pytest.mark.parametrize("with_intensities", [False, True])
def test_merge_frames_matches_cpu_statement(gpu, kitti07, with_intensities):
    """merge_frames_gpu(poses, frames, 0.2) (test_voxelmap.cpp:155-165): same voxel set, means / covariances / intensities
    of the merged cloud equal the CPU statement to f32 storage accuracy"""
    n_frames = 3
    poses = [np.asarray(T, dtype=np.float64) for T in kitti07["poses"][:n_frames]]
    host = []
    rng = np.random.default_rng(5)
    for i in range(n_frames):
        p, c = kitti07[f"points_{i}"], kitti07[f"covs_{i}"]
        it = (rng.integers(0, 128, len(p)) + 128).astype(np.float32) if with_intensities else None
        host.append((p, c, it))
    frames = [gpu.PointCloudGPU(p, c, intensities=it) for p, c, it in host]
    res = 0.2
    merged = gpu.merge_frames_gpu(poses, frames, res, target_points_drop_rate=0.0)
    coords, means, covs, intens = oracle.merge_frames(poses, host, res)
    assert merged.size() == len(coords) and merged.size() < sum(len(p) for p, _, _ in host)
    mp = merged.points_gpu.cpu().numpy().astype(np.float64)
    mc = merged.covs_gpu.cpu().numpy().reshape(-1, 3, 3).transpose(0, 2, 1).astype(np.float64)
    mi = merged.intensities_gpu.cpu().numpy().reshape(-1)
    assert np.isfinite(mp).all() and np.isfinite(mc).all()  # validate_frame
    d, idx = cKDTree(means).query(mp)
    assert d.max() < 2e-5 and len(set(idx.tolist())) == len(coords)  # one merged point per CPU voxel (f32 storage of ~100 m coordinates)
    assert np.abs(mc - covs[idx]).max() < 2e-6
    np.testing.assert_array_equal(mi, intens[idx].astype(np.float32))
    if not with_intensities:
        assert (mi == 0).all()
    # the merged cloud is a usable GPU frame: it can build a voxel map and be a factor source
    vm = gpu.GaussianVoxelMapGPU(1.0)
    vm.insert(merged)
    assert gpu.overlap_gpu(vm, merged) >= 0.99


def test_upload_pack_kernels_are_exact(gpu):
    """add_points_gpu / add_covs_gpu / add_normals_gpu (point_cloud_gpu.cu:110-201): Eigen::Vector{3,4}{d,f} and
    Matrix{3,4}{d,f} host arrays -> float[N][3] / float[N][9] column-major, bit-identical to a host-side cast"""
    rng = np.random.default_rng(11)
    n = 5003
    for dtype in (np.float64, np.float32):
        for dim in (3, 4):
            pts = (rng.normal(size=(n, dim)) * 50.0).astype(dtype)
            nrm = rng.normal(size=(n, dim)).astype(dtype)
            a = rng.normal(size=(n, dim, dim))
            cov = (a @ a.transpose(0, 2, 1)).astype(dtype)
            pc = gpu.PointCloudGPU(pts, cov, normals=nrm)
            assert pc.size() == n
            np.testing.assert_array_equal(pc.points_gpu.cpu().numpy(), pts[:, :3].astype(np.float32))
            np.testing.assert_array_equal(pc.normals_gpu.cpu().numpy(), nrm[:, :3].astype(np.float32))
            want = cov[:, :3, :3].transpose(0, 2, 1).reshape(n, 9).astype(np.float32)  # column-major 3x3
            np.testing.assert_array_equal(pc.covs_gpu.cpu().numpy(), want)
    empty = gpu.PointCloudGPU(np.zeros((0, 3)), np.zeros((0, 3, 3)))
    assert empty.size() == 0
    with pytest.raises(ValueError):
        gpu.PointCloudGPU(np.zeros((4, 5)))


def test_clone_and_times(gpu, kitti00):
    """PointCloudGPU::clone (types/point_cloud_gpu.cu:26-62) and add_times_gpu (:88-105): a deep copy with every attribute, also of a
    frame that lives on the device only; offload / reload keep working on the copy"""
    p, c = kitti00["source_points"], kitti00["source_covs"]
    t = np.linspace(0.0, 0.1, len(p))
    a = gpu.PointCloudGPU(p, c)
    a.add_times(t)
    assert a.memory_usage_gpu() == (12 + 36 + 4) * len(p)
    b = gpu.PointCloudGPU.clone(a)
    assert b.size() == a.size() and b.points_gpu.data_ptr() != a.points_gpu.data_ptr()
    for attr in ("points", "covs", "times"):
        np.testing.assert_array_equal(b.download(attr), a.download(attr))
    np.testing.assert_array_equal(b.download("times"), t.astype(np.float32))
    dev = gpu.PointCloudGPU.from_device(a.points_gpu, a.covs_gpu)
    d = gpu.PointCloudGPU.clone(dev)
    assert d.points_gpu.data_ptr() != a.points_gpu.data_ptr()
    np.testing.assert_array_equal(d.download("covs"), a.download("covs"))
    assert d.offload_gpu() and not d.loaded_on_gpu() and d.reload_gpu()
    np.testing.assert_array_equal(d.download("points"), a.download("points"))
