"""GPU tests of the single-process sharded batch (gp_vgicp_multi_batch_*, gp_multi.hip).  A 1-GPU box rehearses an N-GPU plan
with N shards on one device (host gather); the RCCL path is exercised as a 1-rank communicator.  On a multi-GPU node the
same tests additionally spread the shards over the devices (one shard per device, ncclAllReduce over xGMI)."""
import ctypes as C

import numpy as np
import pytest

from helpers import expmap

pytestmark = pytest.mark.gpu


def _graph(gpu, kitti07, device="cuda:0"):
    poses = kitti07["poses"]
    rng = np.random.default_rng(8191)
    clouds = [gpu.PointCloudGPU(kitti07[f"points_{i}"], kitti07[f"covs_{i}"], device=device) for i in range(5)]
    maps = []
    for c in clouds:
        vm = gpu.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
        vm.insert(c)
        maps.append(vm)
    pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 2), (1, 3), (2, 4), (0, 3), (1, 4), (0, 4)]
    factors = [gpu.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j]) for i, j in pairs]
    values = {k: poses[k] @ expmap(rng.uniform(-0.05, 0.05, 6)) for k in range(5)}
    deltas = [np.linalg.inv(values[i]) @ values[j] for i, j in pairs]
    deltas2 = [d @ expmap(rng.uniform(-0.01, 0.01, 6)) for d in deltas]
    return clouds, maps, factors, deltas, deltas2


def _plain_batch(gpu, factors, deltas, deltas2):
    lib = gpu.load()
    F = len(factors)
    arr = (C.c_void_p * F)(*[f._h.value for f in factors])
    batch, s = C.c_void_p(), C.c_void_p()
    lib.gp_stream_create(C.byref(s))
    gpu._capi.check(lib.gp_vgicp_batch_create(arr, F, s, C.byref(batch)), "batch")
    poses = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in deltas]).copy()
    poses2 = np.stack([np.ascontiguousarray(d.T).reshape(16) for d in deltas2]).copy()
    out, err = np.zeros((F, 122)), np.zeros(F)
    gpu._capi.check(lib.gp_vgicp_batch_linearize(batch, poses.ctypes.data, out.ctypes.data), "linearize")
    gpu._capi.check(lib.gp_vgicp_batch_compute_error(batch, poses.ctypes.data, poses2.ctypes.data, err.ctypes.data), "error")
    lib.gp_vgicp_batch_destroy(batch)
    lib.gp_stream_destroy(s)
    return out, err


def test_sharded_batch_equals_plain_batch(gpu, kitti07):
    """1, 2, 4 shards on one device (contiguous plan from gp_shard_plan) and a round-robin (non-contiguous) assignment: the records
    and the error evaluations are bit-identical to the plain single batch (same kernels, per-factor fixed summation order)"""
    from gtsam_points_amd.distributed import MultiDeviceBatch, partition_factors

    _, _, factors, deltas, deltas2 = _graph(gpu, kitti07)
    ref, ref_err = _plain_batch(gpu, factors, deltas, deltas2)
    assert ref[:, 0].min() > 100  # every factor has inliers
    for shards in [1, 2, 4]:
        parts = partition_factors([int(gpu.load().gp_vgicp_factor_num_points(f._h)) for f in factors], shards)
        shard_of = np.zeros(len(factors), np.int32)
        for k, (b, e) in enumerate(parts):
            shard_of[b:e] = k
        mb = MultiDeviceBatch(factors, shard_of=shard_of, num_shards=shards, use_rccl=0)
        assert mb.num_shards == shards and not mb.uses_rccl
        assert sum(mb.shard_info(k)["num_factors"] for k in range(shards)) == len(factors)
        out = mb.linearize(deltas)
        assert np.array_equal(out, ref), shards
        assert np.array_equal(mb.compute_error(deltas, deltas2), ref_err)
        t = mb.last_timing()
        assert t["ms_compute"] > 0
    rr = MultiDeviceBatch(factors, shard_of=np.arange(len(factors)) % 3, num_shards=3, use_rccl=0)
    assert np.array_equal(rr.linearize(deltas), ref) and np.array_equal(rr.compute_error(deltas, deltas2), ref_err)


def test_rccl_allreduce_path(gpu, kitti07):
    """the ncclAllReduce exchange over the zeroed [F x 122] stack: one shard per visible device (a single GPU is a valid 1-rank
    communicator).  With several devices the factor list is split with gp_shard_plan, clouds and maps are replicated onto the
    shard's device (gp_voxelmap_clone_to_device) and the records must again equal the single-device batch bit for bit."""
    import torch

    from gtsam_points_amd.distributed import MultiDeviceBatch, partition_factors

    clouds, maps, factors, deltas, deltas2 = _graph(gpu, kitti07)
    ref, ref_err = _plain_batch(gpu, factors, deltas, deltas2)
    one = MultiDeviceBatch(factors, use_rccl=1)
    assert one.num_shards == 1 and one.uses_rccl and one.exchange == "all_reduce"
    assert np.array_equal(one.linearize(deltas), ref) and np.array_equal(one.compute_error(deltas, deltas2), ref_err)
    gather = MultiDeviceBatch(factors, use_rccl=2)  # one shard holding everything = one equal contiguous range: the in-place ncclAllGather with one rank
    assert gather.num_shards == 1 and gather.exchange == "all_gather"
    assert np.array_equal(gather.linearize(deltas), ref) and np.array_equal(gather.compute_error(deltas, deltas2), ref_err)
    del gather
    ndev = torch.cuda.device_count()
    if ndev < 2:
        return
    lib = gpu.load()
    pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 2), (1, 3), (2, 4), (0, 3), (1, 4), (0, 4)]
    parts = partition_factors([int(lib.gp_vgicp_factor_num_points(f._h)) for f in factors], ndev)
    sharded, keep = [], []
    for dev, (b, e) in enumerate(parts):
        torch.cuda.set_device(dev)
        gpu._capi.check(lib.gp_set_device(dev), "gp_set_device")
        local_clouds, local_maps = {}, {}
        for (i, j) in pairs[b:e]:
            if j not in local_clouds:
                local_clouds[j] = gpu.PointCloudGPU(kitti07[f"points_{j}"], kitti07[f"covs_{j}"], device=f"cuda:{dev}")
            if i not in local_maps:
                h = C.c_void_p()
                gpu._capi.check(lib.gp_voxelmap_clone_to_device(maps[i]._h, dev, None, C.byref(h)), "clone")
                local_maps[i] = gpu.GaussianVoxelMapGPU(1.0, _handle=h)
            sharded.append(gpu.IntegratedVGICPFactorGPU(i, j, local_maps[i], local_clouds[j]))
        keep.append((local_clouds, local_maps))
    torch.cuda.set_device(0)
    lib.gp_set_device(0)
    for use_rccl in (1, 2, 0):  # all-reduce, all-gather where the plan is equal ranges (else the all-reduce again), no collective
        multi = MultiDeviceBatch(sharded, use_rccl=use_rccl)
        assert multi.num_shards == len({p for p in range(ndev) if parts[p][1] > parts[p][0]}) and multi.uses_rccl == (use_rccl > 0)
        assert np.array_equal(multi.linearize(deltas), ref) and np.array_equal(multi.compute_error(deltas, deltas2), ref_err)
        del multi


def test_voxelmap_clone(gpu, kitti00):
    """gp_voxelmap_clone_to_device: an independent replica (here onto the same device) gives the same linearisation bit for bit"""
    lib = gpu.load()
    tgt = gpu.PointCloudGPU(kitti00["target_points"], kitti00["target_covs"])
    src = gpu.PointCloudGPU(kitti00["source_points"], kitti00["source_covs"])
    vm = gpu.GaussianVoxelMapGPU(0.5, target_points_drop_rate=0.0)
    vm.insert(tgt)
    h = C.c_void_p()
    gpu._capi.check(lib.gp_voxelmap_clone_to_device(vm._h, 0, None, C.byref(h)), "clone")
    clone = gpu.GaussianVoxelMapGPU(0.5, _handle=h)
    assert clone.voxelmap_info.num_voxels == vm.voxelmap_info.num_voxels and lib.gp_voxelmap_has_block_grid(clone._h) == 1
    delta = expmap([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])
    recs = []
    for m in (vm, clone):
        f = gpu.IntegratedVGICPFactorGPU(0, 1, m, src)
        rec = gpu._capi.Linearized6()
        gpu._capi.check(lib.gp_vgicp_factor_linearize(f._h, gpu.types._pose16(delta), C.byref(rec)), "linearize")
        recs.append(bytes(rec))
    assert recs[0] == recs[1]
    del vm  # the clone owns its memory
    f = gpu.IntegratedVGICPFactorGPU(0, 1, clone, src)
    rec = gpu._capi.Linearized6()
    gpu._capi.check(lib.gp_vgicp_factor_linearize(f._h, gpu.types._pose16(delta), C.byref(rec)), "linearize")
    assert bytes(rec) == recs[0]


def test_bench_step_through_rccl_with_one_rank(gpu):
    """bench.py's N > 1 code path (torch.distributed backend "nccl" = RCCL, ShardedLinearizer, all-reduce of the zeroed record stack, the
    sharded C4 leg) with world_size 1: the RCCL / environment failures a gloo rehearsal cannot catch (communicator creation with
    HSA_ENABLE_IPC_MODE_LEGACY=0, device binding, a 1-rank ncclAllReduce on the stream the kernels are issued on) surface here, on a
    1-GPU box.  The record that comes out must match the reference-pinned oracle like the plain synchronous call's."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GP_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import tempfile

    detail = os.path.join(tempfile.mkdtemp(prefix="gp_bench_"), "bench_detail.json")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--source-points", "200000", "--target-points", "400000",
           "--cpu-seconds", "1", "--c4-steps", "2", "--no-configs", "--kernel-iters", "5", "--budget-seconds", "600", "--detail-file", detail]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = p.stdout.rstrip("\n").splitlines()[-1]  # the LAST stdout line is the record (RCCL's printf banner, flushed at exit, must not come behind it)
    assert line.startswith("{"), p.stdout[-600:]
    assert len(line.encode()) < 4096
    r = json.loads(line)
    # the contract's collective by default (VERDICT r05 #2a): RCCL all-reduce; the other forms timed in the same job; every form's stack verified bit for bit
    assert r["n_gpus"] == 1 and r["backend"] == "nccl" and r["rccl_world"] == 1 and r["config"]["exchange"] == "all_reduce"
    assert set(r["exchange_ms"]) == {"all_reduce", "all_gather", "peer"} and all(v is not None and v > 0 for v in r["exchange_ms"].values()), r["exchange_ms"]
    assert r["exchange_verified"] is True
    assert r["parity_ok"] is True and r["parity_inliers_equal"] is True and r["parity_max"] < 1e-6
    assert r["cpu_baseline"] is None  # (the CPU baseline is the N = 1 synchronous run's)
    full = json.load(open(detail))
    assert set(full["exchange_verify_detail"]) == {"all_reduce", "all_gather", "peer"} and all(v["verified"] for v in full["exchange_verify_detail"].values())
    assert "all-reduce" in full["config"]["step"] and "nccl" in full["config"]["exchange_detail"]
    c4 = full["c4"]
    assert c4["factors"] == 4096 and c4["allreduce_ms"] > 0 and 0.3 < c4["inlier_fraction"] < 0.9 and r["legs"]["c4_ms"] == c4["ms_per_linearize"]
    assert c4["exchange"] == "all_reduce"


def test_c4_plan_over_all_visible_devices(gpu):
    """BASELINE configs[3] through the in-library path on REAL devices: the 8-shard plan of a C4 slice (64 submaps' worth would be 512 factors per shard; here 8 factors per
    shard keep the test short) with one shard per visible device, every exchange form -- in-place ncclAllGather, ncclAllReduce, no collective -- against the plain batch on
    device 0, bit for bit.  Skips below two devices (the driver's 8-GPU node runs it; the loop being sharded is cuda/nonlinear_factor_set_gpu.cpp:64-139)."""
    import torch

    from gtsam_points_amd import synthetic
    from gtsam_points_amd.distributed import MultiDeviceBatch, partition_factors

    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs at least two devices")
    lib = gpu.load()
    pairs = synthetic.c4_factor_pairs()[: 8 * ndev]  # equal shards: the all-gather plan
    need = sorted({i for p in pairs for i in p})
    sub = synthetic.make_c4_submaps(need)
    deltas = [synthetic.c4_delta(sub, t, s) for t, s in pairs]

    def build(dev, mine):
        torch.cuda.set_device(dev)
        gpu._capi.check(lib.gp_set_device(dev), "gp_set_device")
        clouds = {i: gpu.PointCloudGPU(sub[i][0], sub[i][1], device=f"cuda:{dev}") for i in sorted({i for p in mine for i in p})}
        maps = {}
        for t in sorted({t for t, _ in mine}):
            maps[t] = gpu.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
            maps[t].insert(clouds[t])
        return [gpu.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s]) for t, s in mine], (clouds, maps)

    plain, keep0 = build(0, pairs)
    ref = MultiDeviceBatch(plain, use_rccl=0).linearize(deltas)
    parts = partition_factors([synthetic.C4_POINTS] * len(pairs), ndev)
    assert [e - b for b, e in parts] == [8] * ndev
    sharded, keep = [], []
    for dev, (b, e) in enumerate(parts):
        fs, k = build(dev, pairs[b:e])
        sharded += fs
        keep.append(k)
    torch.cuda.set_device(0)
    lib.gp_set_device(0)
    for use_rccl, want in [(2, "all_gather"), (1, "all_reduce"), (0, "none"), (-1, "all_gather")]:
        mb = MultiDeviceBatch(sharded, use_rccl=use_rccl)
        assert mb.num_shards == ndev and mb.exchange == want, (mb.exchange, want)
        assert np.array_equal(mb.linearize(deltas), ref), want
        assert np.array_equal(mb.compute_error(deltas, deltas), ref[:, 1]) or np.allclose(mb.compute_error(deltas, deltas), ref[:, 1], rtol=1e-7)
        del mb
