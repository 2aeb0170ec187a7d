"""GPU tests of GaussianVoxelMapGPU against the CPU map (oracle) -- modelled on the reference's
src/test/test_voxelmap.cpp (VoxelMapGPU :211-259, _Intensity :261-299, _IO :301-432)."""
import os

import numpy as np
import pytest

import oracle
from helpers import expmap

pytestmark = pytest.mark.gpu


def _maps(gpu, points, covs, res, intensities=None, **kw):
    cloud = gpu.PointCloudGPU(points, covs, intensities=intensities)
    vm = gpu.GaussianVoxelMapGPU(res, target_points_drop_rate=0.0, **kw)
    vm.insert(cloud)
    om = oracle.OracleVoxelMap(res)
    om.insert(points, covs, intensities)
    return cloud, vm, om


def _reference_lookup(buckets, info, coord):
    """lookup_voxel of cuda/kernels/vector3_hash.cuh:53-76 restated on the downloaded table"""
    M, mask = 0xC6A4A7935BD1E995, (1 << 64) - 1

    def combine(h, k):
        k = (k * M) & mask
        k ^= k >> 47
        k = (k * M) & mask
        h ^= k
        h = (h * M) & mask
        return (h + 0xE6546B64) & mask

    h = 0
    for c in coord:
        h = combine(h, int(c) & mask)
    for i in range(info.max_bucket_scan_count):
        b = buckets[((h + i) & mask) % info.num_buckets]
        if b[3] < 0:
            return -1
        if tuple(b[:3]) == tuple(int(c) for c in coord):
            return int(b[3])
    return -1


@pytest.mark.parametrize("res", [0.5, 1.0, 0.3])
def test_voxel_statistics_match_cpu_map(gpu, kitti00, res):
    _, vm, om = _maps(gpu, kitti00["target_points"], kitti00["target_covs"], res)
    info = vm.voxelmap_info
    assert info.num_voxels == om.num_voxels
    assert abs(info.voxel_resolution - res) < 1e-7 and vm.voxel_resolution() == res
    coords, num_points, means, covs = vm.download_f64()
    oc, on, omean, ocov, _ = om.export()
    order = {tuple(c): i for i, c in enumerate(oc.tolist())}
    idx = np.array([order[tuple(c)] for c in coords.tolist()])
    assert len(set(idx.tolist())) == om.num_voxels  # same voxel set, each exactly once
    np.testing.assert_array_equal(num_points, on[idx])
    assert np.abs(means - omean[idx]).max() < 1e-7 * max(res, 1.0)  # f32 offset from the voxel centre
    assert np.abs(covs - ocov[idx]).max() < 1e-13
    dl = vm.download()
    assert np.isfinite(dl["means"]).all() and np.isfinite(dl["covs"]).all()  # test_voxelmap.cpp:242-250
    assert np.abs(dl["means"] - omean[idx]).max() < 1e-5 and np.abs(dl["covs"] - ocov[idx]).max() < 1e-6
    assert (dl["intensities"] == 0).all()  # no intensities given (test_voxelmap.cpp:271-277)


def test_buckets_unique_in_range_and_findable(gpu, kitti00):
    """test_voxelmap.cpp:352-408: bucket entries unique + in range, every mean findable via its floor coord"""
    _, vm, _ = _maps(gpu, kitti00["target_points"], kitti00["target_covs"], 0.5)
    info = vm.voxelmap_info
    dl = vm.download()
    b = dl["buckets"]
    used = b[b[:, 3] >= 0]
    assert len(used) == info.num_voxels
    assert sorted(used[:, 3].tolist()) == list(range(info.num_voxels))
    assert len({tuple(r[:3]) for r in used.tolist()}) == info.num_voxels
    coords, _, means64, _ = vm.download_f64()
    for v in list(range(0, info.num_voxels, 97)) + [info.num_voxels - 1]:
        c = np.floor(means64[v] / 0.5).astype(int)
        assert tuple(c) == tuple(coords[v])
        assert _reference_lookup(b, info, c) == v


def test_overlap_matches_cpu(gpu, kitti00):
    """self-overlap >= 0.99; |overlap_cpu - overlap_gpu| < 0.01 under a random pose (test_voxelmap.cpp:226-239) --
    here the hit COUNT is identical because both floor in double"""
    _, vm, om = _maps(gpu, kitti00["target_points"], kitti00["target_covs"], 0.5)
    tgt = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])
    src = gpu.PointCloudGPU(kitti00["source_points"], kitti00["source_covs"])
    assert gpu.overlap_gpu(vm, tgt) >= 0.99
    for xi in [np.zeros(6), [0.01, -0.02, 0.015, 0.10, -0.05, 0.03], [0.2, -0.1, 0.3, 1.0, -2.0, 0.5]]:
        T = expmap(xi)
        n = len(kitti00["source_points"])
        assert round(gpu.overlap_gpu(vm, src, T) * n) == round(om.overlap(kitti00["source_points"], T) * n)
    idx = vm.lookup(src, np.eye(4))
    oidx = np.array([om.lookup_coord(np.floor(p.astype(np.float64) / 0.5).astype(int)) for p in kitti00["source_points"][:500]])
    assert ((idx[:500] >= 0) == (oidx >= 0)).all()


def test_intensity_max_semantics(gpu, kitti00):
    """voxel intensity = max of the inserted intensities (atomicMax on the bits, gaussian_voxelmap_gpu.cu:138-139;
    test_voxelmap.cpp:279-298: within [128, 255])"""
    rng = np.random.default_rng(3)
    p, c = kitti00["target_points"][:8000], kitti00["target_covs"][:8000]
    it = rng.uniform(128.0, 255.0, len(p)).astype(np.float32)
    _, vm, om = _maps(gpu, p, c, 1.0, intensities=it)
    coords, _, _, _ = vm.download_f64()
    oc, _, _, _, oint = om.export()
    order = {tuple(k): i for i, k in enumerate(oc.tolist())}
    got = vm.download()["intensities"]
    assert (got >= 128.0).all() and (got <= 255.0).all()
    np.testing.assert_array_equal(got, np.array([oint[order[tuple(k)]] for k in coords.tolist()], dtype=np.float32))


def test_save_load_roundtrip_and_offload(gpu, kitti00, tmp_path):
    """save_compact -> load: same voxel count, means/covs within 1e-3, overlap within 1e-3 (test_voxelmap.cpp:326-431);
    offload_gpu -> device views null -> reload_gpu restores (OffloadableGPU)"""
    _, vm, _ = _maps(gpu, kitti00["target_points"], kitti00["target_covs"], 0.5)
    src = gpu.PointCloudGPU(kitti00["source_points"], kitti00["source_covs"])
    path = os.path.join(tmp_path, "voxelmap.bin")
    vm.save_compact(path)
    header = open(path, "rb").read(200).split(b"\n")
    assert header[0] == b"compact 1" and header[1] == b"resolution 0.5" and header[5] == b"voxel_bytes 56"
    vm2 = gpu.GaussianVoxelMapGPU.load(path)
    assert vm2 is not None and vm2.voxelmap_info.num_voxels == vm.voxelmap_info.num_voxels and vm2.voxel_resolution() == 0.5
    c1, n1, m1, v1 = vm.download_f64()
    c2, n2, m2, v2 = vm2.download_f64()
    o1 = {tuple(c): i for i, c in enumerate(c1.tolist())}
    idx = np.array([o1[tuple(c)] for c in c2.tolist()])
    np.testing.assert_array_equal(n2, n1[idx])
    assert np.abs(m2 - m1[idx]).max() < 1e-3 and np.abs(v2 - v1[idx]).max() < 1e-3
    T = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    assert abs(gpu.overlap_gpu(vm, src, T) - gpu.overlap_gpu(vm2, src, T)) < 1e-3
    assert gpu.GaussianVoxelMapGPU.load(os.path.join(tmp_path, "missing.bin")) is None
    # offload / reload
    before = gpu.overlap_gpu(vm, src, T)
    assert vm.loaded_on_gpu() and vm.memory_usage_gpu() > 0
    assert vm.offload_gpu() and not vm.loaded_on_gpu() and not vm.offload_gpu()
    assert vm.views().buckets is None
    with pytest.raises(gpu.GPError):
        gpu.overlap_gpu(vm, src, T)
    assert vm.reload_gpu() and vm.loaded_on_gpu() and not vm.reload_gpu()
    assert gpu.overlap_gpu(vm, src, T) == before


def test_table_growth_drop_rate_and_empty(gpu, kitti00):
    """tiny initial table: grows by doubling until nothing is dropped; default drop rate 1e-3 may drop <= 0.1 % of
    points (gaussian_voxelmap_gpu.cu:269-291); empty cloud gives an empty map"""
    p, c = kitti00["target_points"], kitti00["target_covs"]
    _, vm, om = _maps(gpu, p, c, 0.5, init_num_buckets=64, max_bucket_scan_count=4)
    info = vm.voxelmap_info
    assert info.num_voxels == om.num_voxels and info.num_buckets >= info.num_voxels and info.num_buckets % 64 == 0
    _, vm3, _ = _maps(gpu, p, c, 0.5, init_num_buckets=1000)  # non power-of-two table: modulo path
    assert vm3.voxelmap_info.num_voxels == om.num_voxels and vm3.voxelmap_info.num_buckets % 1000 == 0
    cloud = gpu.PointCloudGPU(p, c)
    vmd = gpu.GaussianVoxelMapGPU(0.5, init_num_buckets=8192, target_points_drop_rate=1e-3)
    vmd.insert(cloud)
    kept = vmd.download()["num_points"].sum()
    assert kept >= (1 - 1e-3) * len(p) and vmd.voxelmap_info.num_voxels <= om.num_voxels
    empty = gpu.PointCloudGPU(np.zeros((1, 3), np.float32), np.zeros((1, 3, 3), np.float32))
    empty.num_points = 0
    vme = gpu.GaussianVoxelMapGPU(0.5)
    vme.insert(empty)
    assert vme.voxelmap_info.num_voxels == 0
    assert gpu.overlap_gpu(vme, cloud) == 0.0
    with pytest.raises(gpu.GPError):
        vme.insert(gpu.PointCloudGPU(p))  # no covs: the reference abort()s (gaussian_voxelmap_gpu.cu:212-215)


def test_file_interop_with_the_reference_cpu_map(gpu, kitti00, tmp_path):
    """src/test/test_voxelmap.cpp:301-432 across implementations: a file written by gp_voxelmap_save_compact is loaded by the
    REFERENCE's own GaussianVoxelMapCPU::load (oracle/_ref/libref.so, compiled from gaussian_voxelmap_cpu.cpp where it lies) and a
    file written by the reference's GaussianVoxelMapCPU::save_compact is loaded by gp_voxelmap_load; voxel sets identical,
    means / covs within 1e-3 (the on-disk record is float), overlap within 1e-3 -- the reference's own gates (:326-408, :418-431)"""
    from oracle import refcapi

    if not refcapi.available():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference at build time)")
    intens = np.linalg.norm(kitti00["target_points"], axis=1).astype(np.float32)
    cloud, vm, _ = _maps(gpu, kitti00["target_points"], kitti00["target_covs"], 0.5, intensities=intens)
    ref = refcapi.RefVoxelMap(0.5)
    ref.insert(kitti00["target_points"], kitti00["target_covs"])
    src = kitti00["source_points"]
    delta = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    src_gpu = gpu.PointCloudGPU(src)

    def compare(gpu_map, ref_map, tol):
        coords, npts, means, covs = gpu_map.download_f64()
        rc, rn, rmean, rcov, _ = ref_map.export()
        assert len(coords) == len(rc) == vm.voxelmap_info.num_voxels
        order = {tuple(c): i for i, c in enumerate(rc.tolist())}
        idx = np.array([order[tuple(c)] for c in coords.tolist()])  # KeyError = a voxel the other side does not have
        assert len(set(idx.tolist())) == len(rc)
        np.testing.assert_array_equal(npts, rn[idx])
        assert np.abs(means - rmean[idx]).max() < tol and np.abs(covs - rcov[idx]).max() < tol
        for T in (np.eye(4), delta):
            assert abs(gpu.overlap_gpu(gpu_map, src_gpu, T) - ref_map.overlap(src, T)) < 1e-3

    compare(vm, ref, 1e-6)  # before any file: same statistics (f64 vs f64)
    # GPU -> file -> reference CPU map
    p1 = str(tmp_path / "from_gpu.bin")
    vm.save_compact(p1)
    ref_loaded = refcapi.RefVoxelMap.load(p1)
    assert ref_loaded is not None and ref_loaded.num_voxels == vm.voxelmap_info.num_voxels and abs(ref_loaded.voxel_resolution() - 0.5) < 1e-12
    compare(vm, ref_loaded, 1e-3)
    _, _, _, _, ri = ref_loaded.export()
    dl = vm.download()
    coords = vm.download_f64()[0]
    order = {tuple(c): i for i, c in enumerate(ref_loaded.export()[0].tolist())}
    idx = np.array([order[tuple(c)] for c in coords.tolist()])
    assert np.abs(dl["intensities"] - ri[idx]).max() < 1e-3  # the max-intensity field survives too
    # reference CPU map -> file -> GPU
    p2 = str(tmp_path / "from_ref.bin")
    ref.save_compact(p2)
    gpu_loaded = gpu.GaussianVoxelMapGPU.load(p2)
    assert gpu_loaded.voxelmap_info.num_voxels == ref.num_voxels and abs(gpu_loaded.voxel_resolution() - 0.5) < 1e-12
    compare(gpu_loaded, ref, 1e-3)
    # and a VGICP factor on the map loaded from the reference's file agrees with the factor on the GPU-built map to the file's precision
    import ctypes as C

    srcc = gpu.PointCloudGPU(kitti00["source_points"], kitti00["source_covs"])
    recs = []
    for m in (vm, gpu_loaded):
        f = gpu.IntegratedVGICPFactorGPU(0, 1, m, srcc)
        rec = gpu._capi.Linearized6()
        gpu._capi.check(f._lib.gp_vgicp_factor_linearize(f._h, gpu.types._pose16(delta), C.byref(rec)), "linearize")
        recs.append(gpu.LinearizedSystem6(rec))
    assert recs[0].num_inliers == recs[1].num_inliers
    assert np.linalg.norm(recs[0].H_source - recs[1].H_source) / np.linalg.norm(recs[0].H_source) < 1e-3
    del cloud


def test_binned_build_is_bit_reproducible_and_matches_the_hashed_build(gpu, kitti00):
    """the default (binned) build sorts the points by voxel with a stable radix sort and sums every voxel in ascending point order:
    two builds of the same cloud are bit-identical (records, reference-visible arrays, voxel numbering).  The reference-shaped
    hashed build (atomicCAS claims + atomic f64 sums; also the fallback for huge bounding boxes) gives the same voxel set and the
    same statistics up to the summation order."""
    lib = gpu.load()
    intens = np.abs(kitti00["target_points"][:, 0]).astype(np.float32)
    cloud = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"], intensities=intens)

    def build(hashed=False):
        vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
        if hashed:
            gpu._capi.check(lib.gp_voxelmap_set_tuning(vm._h, gpu._capi.GP_TUNE_MAP_BUILD, 1), "gp_voxelmap_set_tuning")
        vm.insert(cloud)
        return vm

    a, b = build(), build()
    ca, na, ma, va = a.download_f64()
    cb, nb, mb, vb = b.download_f64()
    assert np.array_equal(ca, cb) and np.array_equal(na, nb) and np.array_equal(ma, mb) and np.array_equal(va, vb)
    da, db = a.download(), b.download()
    for k in ["num_points", "means", "covs", "intensities"]:
        assert np.array_equal(da[k], db[k]), k
    assert lib.gp_voxelmap_has_block_grid(a._h) == 1
    h = build(hashed=True)  # gp_voxelmap_set_tuning(GP_TUNE_MAP_BUILD, 1) on THIS map
    ch, nh, mh, vh = h.download_f64()
    assert len(ch) == len(ca)
    order = {tuple(c): i for i, c in enumerate(ch.tolist())}
    idx = np.array([order[tuple(c)] for c in ca.tolist()])
    assert np.array_equal(na, nh[idx]) and np.abs(ma - mh[idx]).max() < 1e-7 and np.abs(va - vh[idx]).max() < 1e-13
    assert np.array_equal(da["intensities"], h.download()["intensities"][idx])


def test_build_survives_a_sort_that_gives_up(gpu, kitti00):
    """the builds' radix sort waits on tiles with smaller indices, which rests on the device starting workgroups in blockIdx order; a sort that finds a
    predecessor missing for too long raises a fault word instead of spinning for ever, and the build runs again through the one-class sort.  The test hook makes the
    first sort of the next build report exactly that: the map and the k-NN covariances built through the fallback equal the ordinary ones bit for bit."""
    import torch

    from gtsam_points_amd.features import estimate_covariances_gpu

    lib = gpu.load()
    cloud = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])

    def build():
        vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
        vm.insert(cloud)
        return vm

    a = build()
    before = lib.gp_debug_sort_fallbacks()
    gpu._capi.check(lib.gp_debug_inject_sort_fault(1), "inject")
    b = build()
    assert lib.gp_debug_sort_fallbacks() == before + 1
    c = build()  # the hook is spent
    assert lib.gp_debug_sort_fallbacks() == before + 1
    ref = a.download_f64()
    for other in (b, c):
        for x, y in zip(ref, other.download_f64()):
            assert np.array_equal(x, y)
    q = gpu.PointCloudGPU(kitti00["source_points"])
    estimate_covariances_gpu(q, k_neighbors=10)
    want = q.covs_gpu.clone()
    gpu._capi.check(lib.gp_debug_inject_sort_fault(1), "inject")
    estimate_covariances_gpu(q, k_neighbors=10)
    torch.cuda.synchronize()
    assert lib.gp_debug_sort_fallbacks() == before + 2
    assert torch.equal(want, q.covs_gpu)
    # a voided sort leaves HOLES, i.e. arbitrary keys: the hook's negative form also writes keys far outside every cell range into the first tile's part of
    # the output.  The cell kernel must not index anything with them (it did: blocks[key >> 6]); the rebuilt map is the same map, bit for bit, and so are the covariances
    gpu._capi.check(lib.gp_debug_inject_sort_fault(-1), "inject (corrupting)")
    d = build()
    assert lib.gp_debug_sort_fallbacks() == before + 3
    for x, y in zip(ref, d.download_f64()):
        assert np.array_equal(x, y)
    gpu._capi.check(lib.gp_debug_inject_sort_fault(-1), "inject (corrupting)")
    estimate_covariances_gpu(q, k_neighbors=10)
    torch.cuda.synchronize()
    assert lib.gp_debug_sort_fallbacks() == before + 4
    assert torch.equal(want, q.covs_gpu)
    e = build()  # (and nothing is left behind: the next build is the fast path again)
    assert lib.gp_debug_sort_fallbacks() == before + 4
    for x, y in zip(ref, e.download_f64()):
        assert np.array_equal(x, y)


def test_non_finite_points_are_skipped(gpu, kitti00):
    """LiDAR clouds contain NaN / inf returns: they belong to no voxel (the reference floors them into undefined coordinates)"""
    p = kitti00["target_points"].copy()
    c = kitti00["target_covs"]
    bad = [5, 100, 4096, len(p) - 1]
    p[bad[0]] = np.nan
    p[bad[1], 1] = np.inf
    p[bad[2], 2] = -np.inf
    p[bad[3], 0] = np.nan
    keep = np.ones(len(p), bool)
    keep[bad] = False
    _, vm, _ = _maps(gpu, p, c, 0.5)
    _, _, om = _maps(gpu, p[keep], c[keep], 0.5)
    assert vm.voxelmap_info.num_voxels == om.num_voxels
    coords, num_points, means, covs = vm.download_f64()
    oc, on, omean, ocov, _ = om.export()
    order = {tuple(x): i for i, x in enumerate(oc.tolist())}
    idx = np.array([order[tuple(x)] for x in coords.tolist()])
    np.testing.assert_array_equal(num_points, on[idx])
    assert np.abs(covs - ocov[idx]).max() < 1e-13 and num_points.sum() == keep.sum()


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 255, 4095, 4096, 4097, 8193])
def test_binned_build_at_sizes_around_the_tile_boundaries(gpu, kitti00, n):
    """round 4: the build's kernels work on tiles of 4096 (keys, sort passes, cells) and batches of 512 rows (statistics): clouds of 1 point, one tile minus / plus one
    point, two tiles plus one -- with a non-finite point at the very end and at the very start -- against the CPU map"""
    p = kitti00["target_points"][:n].copy()
    c = kitti00["target_covs"][:n]
    for bad in ([], [n - 1], [0]):
        q = p.copy()
        keep = np.ones(n, bool)
        for b in bad:
            q[b, 0] = np.nan
            keep[b] = False
        if keep.sum() == 0:
            continue
        _, vm, _ = _maps(gpu, q, c, 0.5)
        _, _, om = _maps(gpu, q[keep], c[keep], 0.5)
        assert vm.voxelmap_info.num_voxels == om.num_voxels
        coords, num_points, means, covs = vm.download_f64()
        oc, on, omean, ocov, _ = om.export()
        order = {tuple(x): i for i, x in enumerate(oc.tolist())}
        idx = np.array([order[tuple(x)] for x in coords.tolist()])
        np.testing.assert_array_equal(num_points, on[idx])
        assert num_points.sum() == keep.sum()
        assert np.abs(means - omean[idx]).max() < 1e-7 and np.abs(covs - ocov[idx]).max() < 1e-13


def test_one_voxel_with_many_points_and_a_cloud_in_one_cell(gpu):
    """statistics batches: a voxel far larger than the 512-row batch, next to tiny ones; and every point in ONE voxel"""
    rng = np.random.default_rng(3)
    big = rng.uniform(0.01, 0.49, size=(5000, 3)).astype(np.float32)  # one voxel at 0.5 m
    small = (rng.uniform(0, 0.49, size=(40, 3)) + np.arange(40)[:, None] * 3.0 + 2.0).astype(np.float32)
    for pts in (np.concatenate([small[:20], big, small[20:]]), big):
        covs = np.tile(np.diag([1e-3, 1.0, 2.0]).astype(np.float32), (len(pts), 1, 1))
        _, vm, om = _maps(gpu, pts, covs, 0.5)
        assert vm.voxelmap_info.num_voxels == om.num_voxels
        coords, num_points, means, covs64 = vm.download_f64()
        oc, on, omean, ocov, _ = om.export()
        order = {tuple(x): i for i, x in enumerate(oc.tolist())}
        idx = np.array([order[tuple(x)] for x in coords.tolist()])
        np.testing.assert_array_equal(num_points, on[idx])
        assert np.abs(means - omean[idx]).max() < 1e-6 and np.abs(covs64 - ocov[idx]).max() < 1e-12


def test_bucket_load_factor_is_a_tunable_of_the_map(gpu, kitti00):
    """the binned build enters the reference's bucket-table doubling sequence (gaussian_voxelmap_gpu.cu:269-291) at the first size that holds the voxels at
    GP_TUNE_BUCKET_LOAD per cent (default 33); info.num_buckets says what was taken, and the lookups through the reference-visible table still find every voxel."""
    lib = gpu.load()
    cloud = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])
    sizes, overlaps = {}, {}
    for load in (33, 67, 10):
        vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
        gpu._capi.check(lib.gp_voxelmap_set_tuning(vm._h, gpu._capi.GP_TUNE_BUCKET_LOAD, load), "gp_voxelmap_set_tuning")
        vm.insert(cloud)
        info = vm.voxelmap_info
        sizes[load] = info.num_buckets
        assert info.num_buckets * load >= 100 * info.num_voxels
        b = vm.download()["buckets"]
        used = b[b[:, 3] >= 0]
        assert len(b) == info.num_buckets and sorted(used[:, 3].tolist()) == list(range(info.num_voxels))  # every voxel sits in exactly one bucket (test_voxelmap.cpp:245,272)
        overlaps[load] = len(used)
    assert sizes[10] >= sizes[33] >= sizes[67]
    vm = gpu.GaussianVoxelMapGPU(0.5)
    assert lib.gp_voxelmap_set_tuning(vm._h, gpu._capi.GP_TUNE_BUCKET_LOAD, 3) != 0
