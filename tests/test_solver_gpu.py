"""GPU tests of the device-side normal equations: DenseLinearSystemBuilder + buildDampedSystem +
DenseLinearSolver::solve restated with numpy on the host as the checker (the reference's own solve is GTSAM / Eigen, absent)."""
import numpy as np
import pytest

from helpers import expmap, pose_error

pytestmark = pytest.mark.gpu


def _graph(gpu, kitti07, res=1.0):
    n = 5
    clouds = [gpu.PointCloudGPU(kitti07[f"points_{i}"], kitti07[f"covs_{i}"]) for i in range(n)]
    maps = []
    for c in clouds:
        vm = gpu.GaussianVoxelMapGPU(res, target_points_drop_rate=0.0)
        vm.insert(c)
        maps.append(vm)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n) if j - i <= 2]  # 7 binary factors
    factors = [gpu.IntegratedVGICPFactorGPU(i, j, maps[i], clouds[j]) for i, j in pairs]
    rng = np.random.default_rng(8191)
    values = {i: np.asarray(kitti07["poses"][i], dtype=np.float64) @ expmap(rng.uniform(-0.02, 0.02, 6)) for i in range(n)}
    return clouds, maps, pairs, factors, values


def _host_system(records, slots, n_slots):
    """DenseLinearSystemBuilder (linear_system_builder.cpp:39-48) on the host: scatter the Hessian blocks by key"""
    A, b, c = np.zeros((6 * n_slots, 6 * n_slots)), np.zeros(6 * n_slots), 0.0
    for rec, (st, ss) in zip(records, slots):
        Ht, Hs, Hts = rec[2:38].reshape(6, 6).T, rec[38:74].reshape(6, 6).T, rec[74:110].reshape(6, 6).T
        bt, bs = rec[110:116], rec[116:122]
        c += rec[1]
        if st >= 0:
            A[6 * st : 6 * st + 6, 6 * st : 6 * st + 6] += Ht
            b[6 * st : 6 * st + 6] -= bt
        if ss >= 0:
            A[6 * ss : 6 * ss + 6, 6 * ss : 6 * ss + 6] += Hs
            b[6 * ss : 6 * ss + 6] -= bs
        if st >= 0 and ss >= 0:
            A[6 * st : 6 * st + 6, 6 * ss : 6 * ss + 6] += Hts
            A[6 * ss : 6 * ss + 6, 6 * st : 6 * st + 6] += Hts.T
    return A, b, c


def test_build_damp_solve_match_host(gpu, kitti07):
    _, _, pairs, factors, values = _graph(gpu, kitti07)
    rec_dev = gpu.linearize_on_device(factors, values)
    rec = rec_dev.cpu().numpy()
    # pose 0 fixed (no slot), poses 1..4 -> slots 0..3
    slots = [(i - 1, j - 1) for i, j in pairs]
    sys = gpu.DenseLinearSystemGPU(4, slots)
    A, b, c = sys.build(rec_dev).download()
    Ah, bh, ch = _host_system(rec, slots, 4)
    assert np.abs(A - Ah).max() <= 1e-12 * np.abs(Ah).max() and np.abs(b - bh).max() <= 1e-12 * np.abs(bh).max() and abs(c - ch) <= 1e-12 * abs(ch)
    assert np.array_equal(A, A.T)
    for lam, diag in [(0.0, False), (1e-3, False), (10.0, True)]:
        sys.build(rec_dev, lam=lam, diagonal_damping=diag)
        Ad = sys.download()[0]
        want = Ah + (lam * np.diag(np.clip(np.diag(Ah), 1e-6, 1e32)) if diag else lam * np.eye(24))
        assert np.abs(Ad - want).max() <= 1e-12 * np.abs(want).max()
        x = sys.solve()
        xh = np.linalg.solve(want, bh)
        assert np.linalg.norm(x - xh) <= 1e-9 * np.linalg.norm(xh), (lam, diag)
    # deterministic: same records -> bit-identical solution
    x1 = sys.build(rec_dev, lam=1e-3).solve()
    x2 = sys.build(rec_dev, lam=1e-3).solve()
    assert np.array_equal(x1, x2)
    # all five poses free and no prior: the gauge freedom makes the system singular -> reported, not swallowed
    free = gpu.DenseLinearSystemGPU(5, pairs)
    with pytest.raises(gpu.GPError):
        free.build(rec_dev).solve()
    # ... a prior on pose 0 fixes it
    prior = np.zeros(30)
    prior[:6] = 1e6
    x = free.build(rec_dev, prior_diag=prior).solve()
    Af, bf, _ = _host_system(rec, pairs, 5)
    xf = np.linalg.solve(Af + np.diag(prior), bf)
    assert np.linalg.norm(x - xf) <= 1e-8 * np.linalg.norm(xf)


def test_larger_random_spd_system(gpu):
    """the blocked Cholesky on a synthetic 60-pose chain-with-loops graph (records made on the host, uploaded)"""
    import torch

    rng = np.random.default_rng(3)
    P, pairs = 60, []
    for i in range(P):
        for d in (1, 2, 7):
            if i + d < P:
                pairs.append((i, i + d))
    rec = np.zeros((len(pairs), 122))
    for k in range(len(pairs)):
        J = rng.normal(size=(40, 12))
        H = J.T @ J  # PSD 12x12: [[Ht, Hts], [Hts^T, Hs]]
        rec[k, 0], rec[k, 1] = 40, rng.uniform(1, 2)
        rec[k, 2:38], rec[k, 38:74], rec[k, 74:110] = H[:6, :6].T.reshape(36), H[6:, 6:].T.reshape(36), H[:6, 6:].T.reshape(36)
        rec[k, 110:122] = rng.normal(size=12)
    rec_dev = torch.from_numpy(rec).cuda()
    sys = gpu.DenseLinearSystemGPU(P, pairs)
    x = sys.build(rec_dev, lam=1e-2).solve()
    Ah, bh, _ = _host_system(rec, pairs, P)
    xh = np.linalg.solve(Ah + 1e-2 * np.eye(6 * P), bh)
    assert np.linalg.norm(x - xh) <= 1e-9 * np.linalg.norm(xh)


def test_unary_only_and_single_pose_systems(gpu):
    """factors whose target is fixed contribute H_source / b_source only (the unary HessianFactor of
    integrated_vgicp_factor_gpu.cpp:210-213); a one-pose system is a single 6x6 solve"""
    import torch

    rng = np.random.default_rng(9)
    rec = np.zeros((5, 122))
    for k in range(5):
        J = rng.normal(size=(30, 12))
        H = J.T @ J
        rec[k, 0], rec[k, 1] = 30, 1.0 + k
        rec[k, 2:38], rec[k, 38:74], rec[k, 74:110] = H[:6, :6].T.reshape(36), H[6:, 6:].T.reshape(36), H[:6, 6:].T.reshape(36)
        rec[k, 110:122] = rng.normal(size=12)
    rec_dev = torch.from_numpy(rec).cuda()
    slots = [(-1, 0), (-1, 1), (-1, 2), (-1, 0), (-1, 2)]
    A, b, c = gpu.DenseLinearSystemGPU(3, slots).build(rec_dev).download()
    Ah, bh, ch = _host_system(rec, slots, 3)
    assert np.allclose(A, Ah, rtol=1e-13, atol=0) and np.allclose(b, bh, rtol=1e-13) and abs(c - ch) < 1e-12
    assert np.abs(A[:6, 6:]).max() == 0.0  # no coupling between poses
    x = gpu.DenseLinearSystemGPU(3, slots).build(rec_dev).solve()
    assert np.linalg.norm(x - np.linalg.solve(Ah, bh)) <= 1e-10 * np.linalg.norm(x)
    one = gpu.DenseLinearSystemGPU(1, [(-1, 0)])
    x1 = one.build(rec_dev[:1].contiguous()).solve()
    A1, b1, _ = _host_system(rec[:1], [(-1, 0)], 1)
    assert np.linalg.norm(x1 - np.linalg.solve(A1, b1)) <= 1e-11 * np.linalg.norm(x1)
    with pytest.raises(gpu.GPError):
        gpu.DenseLinearSystemGPU(2, [(0, 0)])  # a factor needs two different poses
    with pytest.raises(gpu.GPError):
        gpu.DenseLinearSystemGPU(2, [(0, 2)])  # slot out of range


def test_lm_with_device_solve_reaches_the_alignment_gate(gpu, kitti07):
    """the reference's alignment gate (test_matching_cost_factors.cpp:227: < 0.015 rad / 0.15 m) with every linear algebra
    step of the LM loop on the GPU: batched linearise -> records in HBM -> device assembly + damping + Cholesky"""
    clouds, maps, pairs, factors, values0 = _graph(gpu, kitti07)
    slots = [(i - 1, j - 1) for i, j in pairs]
    sys = gpu.DenseLinearSystemGPU(4, slots)
    fset = gpu.NonlinearFactorSetGPU()
    for f in factors:
        fset.add(f)
    values, lam = dict(values0), 1e-5
    fset.linearize(values)
    err = sum(f.error(values) for f in factors)
    for _ in range(30):
        rec_dev = gpu.linearize_on_device(factors, values)
        improved = False
        for _try in range(12):
            dx = sys.build(rec_dev, lam=lam, diagonal_damping=True).solve()
            new_values = dict(values)
            for k in range(1, 5):
                new_values[k] = values[k] @ expmap(dx[6 * (k - 1) : 6 * k])
            fset.linearize(values)  # error() evaluates with the correspondences frozen at the linearisation point
            fset.error(new_values)
            new_err = sum(f.error(new_values) for f in factors)
            if new_err < err:
                improved, lam = True, max(lam / 10.0, 1e-12)
                break
            lam *= 10.0
        if not improved:
            break
        rel = (err - new_err) / max(err, 1e-300)
        values, err = new_values, new_err
        if rel < 1e-4:
            break
    gt = [np.asarray(T, dtype=np.float64) for T in kitti07["poses"]]
    for k in range(1, 5):
        ang, trans = pose_error(np.linalg.inv(values[0]) @ values[k], np.linalg.inv(gt[0]) @ gt[k])
        assert ang < 0.015 and trans < 0.15, (k, ang, trans)


# ---- block-sparse LL^T over the pose graph (gp_sparse.hip): the dense path and numpy are the checkers ---------------------------


def _random_records(pairs, rng, rows=40):
    rec = np.zeros((len(pairs), 122))
    for k in range(len(pairs)):
        J = rng.normal(size=(rows, 12))
        H = J.T @ J  # PSD 12x12: [[Ht, Hts], [Hts^T, Hs]]
        rec[k, 0], rec[k, 1] = rows, rng.uniform(1, 2)
        rec[k, 2:38], rec[k, 38:74], rec[k, 74:110] = H[:6, :6].T.reshape(36), H[6:, 6:].T.reshape(36), H[:6, 6:].T.reshape(36)
        rec[k, 110:122] = rng.normal(size=12)
    return rec


@pytest.mark.parametrize("ordering", ["natural", "nd", "amd", "amd1", "auto"])
def test_sparse_system_matches_dense_and_numpy(gpu, kitti07, ordering):
    _, _, pairs, factors, values = _graph(gpu, kitti07)
    rec_dev = gpu.linearize_on_device(factors, values)
    rec = rec_dev.cpu().numpy()
    slots = [(i - 1, j - 1) for i, j in pairs]  # pose 0 fixed
    sp = gpu.SparseLinearSystemGPU(4, slots, ordering=ordering)
    A, b, c = sp.build(rec_dev).download()
    Ah, bh, ch = _host_system(rec, slots, 4)
    assert np.abs(A - Ah).max() <= 1e-12 * np.abs(Ah).max() and np.abs(b - bh).max() <= 1e-12 * np.abs(bh).max() and abs(c - ch) <= 1e-12 * abs(ch)
    for lam, diag in [(0.0, False), (1e-3, False), (10.0, True)]:
        Ad = sp.build(rec_dev, lam=lam, diagonal_damping=diag).download()[0]
        want = Ah + (lam * np.diag(np.clip(np.diag(Ah), 1e-6, 1e32)) if diag else lam * np.eye(24))
        assert np.abs(Ad - want).max() <= 1e-12 * np.abs(want).max()
        x = sp.build(rec_dev, lam=lam, diagonal_damping=diag).solve()
        xd = gpu.DenseLinearSystemGPU(4, slots).build(rec_dev, lam=lam, diagonal_damping=diag).solve()
        xh = np.linalg.solve(want, bh)
        assert np.linalg.norm(x - xh) <= 1e-9 * np.linalg.norm(xh), (lam, diag)
        assert np.linalg.norm(x - xd) <= 1e-9 * np.linalg.norm(xd), (lam, diag)
    # deterministic
    assert np.array_equal(sp.build(rec_dev, lam=1e-3).solve(), sp.build(rec_dev, lam=1e-3).solve())
    # gauge freedom -> indeterminate, reported; a prior fixes it
    free = gpu.SparseLinearSystemGPU(5, pairs, ordering=ordering)
    with pytest.raises(gpu.GPError):
        free.build(rec_dev).solve()
    prior = np.zeros(30)
    prior[:6] = 1e6
    x = free.build(rec_dev, prior_diag=prior).solve()
    Af, bf, _ = _host_system(rec, pairs, 5)
    xf = np.linalg.solve(Af + np.diag(prior), bf)
    assert np.linalg.norm(x - xf) <= 1e-8 * np.linalg.norm(xf)


@pytest.mark.parametrize("ordering", ["natural", "nd", "amd", "amd1", "auto"])
@pytest.mark.parametrize("case", ["chain512", "loops300", "grid", "band512", "unary+isolated"])
def test_sparse_solve_on_synthetic_graphs(gpu, case, ordering):
    import torch

    rng = np.random.default_rng(11)
    if case == "chain512":
        P = 512
        pairs = [(-1, 0)] + [(i, i + 1) for i in range(P - 1)]
    elif case == "loops300":
        P = 300
        pairs = [(-1, 0)] + [(i, i + d) for i in range(P) for d in (1, 2) if i + d < P] + [(int(a), int(b)) for a, b in rng.integers(0, P, (40, 2)) if a != b]
    elif case == "grid":
        w = 14
        P = w * w
        pairs = [(r * w + c, r * w + c + 1) for r in range(w) for c in range(w - 1)] + [(r * w + c, (r + 1) * w + c) for r in range(w - 1) for c in range(w)]
    elif case == "band512":  # i -> i + 1, i + 2, i + 7: wide separators, the graph the single-workgroup top of round 2 spent 4-5 ms on
        P = 512
        pairs = [(-1, 0)] + [(i, i + d) for i in range(P) for d in (1, 2, 7) if i + d < P]
    else:
        P = 12  # two separate chains, poses tied to fixed ones, and nothing but a unary factor on pose 11
        pairs = [(-1, 0), (0, 1), (1, 2), (-1, 5), (5, 6), (6, 7), (7, 5), (-1, 11), (-1, 3), (3, 4), (-1, 8), (8, 9), (9, 10)]
    rec = _random_records(pairs, rng)
    rec_dev = torch.from_numpy(rec).cuda()
    sp = gpu.SparseLinearSystemGPU(P, pairs, ordering=ordering)
    info = sp.info()
    assert info["nnz_l_blocks"] >= info["nnz_a_blocks"] >= P
    x = sp.build(rec_dev, lam=1e-2).solve()
    Ah, bh, _ = _host_system(rec, pairs, P)
    xh = np.linalg.solve(Ah + 1e-2 * np.eye(6 * P), bh)
    assert np.linalg.norm(x - xh) <= 1e-9 * np.linalg.norm(xh), info
    A = sp.build(rec_dev, lam=1e-2).download()[0]
    assert np.abs(A - (Ah + 1e-2 * np.eye(6 * P))).max() <= 1e-12 * np.abs(Ah).max()


def test_sparse_lm_reaches_the_alignment_gate(gpu, kitti07):
    """the LM loop of test_lm_with_device_solve_reaches_the_alignment_gate with the block-sparse solver"""
    clouds, maps, pairs, factors, values0 = _graph(gpu, kitti07)
    slots = [(i - 1, j - 1) for i, j in pairs]
    sys = gpu.SparseLinearSystemGPU(4, slots, ordering="nd")
    fset = gpu.NonlinearFactorSetGPU()
    for f in factors:
        fset.add(f)
    values, lam = dict(values0), 1e-5
    fset.linearize(values)
    err = sum(f.error(values) for f in factors)
    for _ in range(30):
        rec_dev = gpu.linearize_on_device(factors, values)
        improved = False
        for _try in range(12):
            dx = sys.build(rec_dev, lam=lam, diagonal_damping=True).solve()
            new_values = dict(values)
            for k in range(1, 5):
                new_values[k] = values[k] @ expmap(dx[6 * (k - 1) : 6 * k])
            fset.linearize(values)
            fset.error(new_values)
            new_err = sum(f.error(new_values) for f in factors)
            if new_err < err:
                improved, lam = True, max(lam / 10.0, 1e-12)
                break
            lam *= 10.0
        if not improved:
            break
        rel = (err - new_err) / max(err, 1e-300)
        values, err = new_values, new_err
        if rel < 1e-4:
            break
    gt = [np.asarray(T, dtype=np.float64) for T in kitti07["poses"]]
    for k in range(1, 5):
        ang, trans = pose_error(np.linalg.inv(values[0]) @ values[k], np.linalg.inv(gt[0]) @ gt[k])
        assert ang < 0.015 and trans < 0.15, (k, ang, trans)


@pytest.mark.parametrize("kind", ["dense", "sparse-nd", "sparse-amd1", "sparse-natural"])
def test_step_is_build_download_solve_in_one_call(gpu, kind):
    """gp_*_system_step (the optimizer's tryLambda, levenberg_marquardt_ext.cpp:188-260, as one call with one wait) returns the same x, b, c bit for bit as
    build -> download -> solve, with every damping form and a prior; an indeterminate system is reported and still hands over b and c"""
    import torch

    rng = np.random.default_rng(5)
    P = 40
    pairs = [(-1, 0)] + [(i, i + d) for i in range(P) for d in (1, 2, 5) if i + d < P]
    rec = _random_records(pairs, rng)
    rec_dev = torch.from_numpy(rec).cuda()

    def make():
        return gpu.DenseLinearSystemGPU(P, pairs) if kind == "dense" else gpu.SparseLinearSystemGPU(P, pairs, ordering=kind.split("-")[1])

    three, one = make(), make()
    prior = rng.uniform(0.0, 2.0, 6 * P)
    for lam, diag, pr in [(0.0, False, None), (1e-3, False, None), (10.0, True, None), (1e-2, False, prior), (0.5, True, prior)]:
        three.build(rec_dev, lam=lam, diagonal_damping=diag, prior_diag=pr)
        _, b3, c3 = three.download()
        x3 = three.solve()
        x1, b1, c1 = one.step(rec_dev, lam=lam, diagonal_damping=diag, prior_diag=pr)
        assert np.array_equal(x1, x3) and np.array_equal(b1, b3) and c1 == c3, (lam, diag, pr is not None)
    # the caller's arrays are filled in place, call after call
    out = (np.zeros(6 * P), np.zeros(6 * P), np.zeros(1))
    for lam in (1e-3, 1e-1):
        x1, b1, c1 = one.step(rec_dev, lam=lam, out=out)
        assert x1 is out[0] and np.array_equal(x1, three.build(rec_dev, lam=lam).solve())
    # gauge freedom: indeterminate, reported; b and c arrive all the same
    free_pairs = [(i, i + 1) for i in range(P - 1)]
    rec_f = _random_records(free_pairs, rng)
    rec_f[:, 2:110] *= 0.0  # no curvature at all: the first pivot is zero
    rec_f_dev = torch.from_numpy(rec_f).cuda()
    free = gpu.DenseLinearSystemGPU(P, free_pairs) if kind == "dense" else gpu.SparseLinearSystemGPU(P, free_pairs, ordering=kind.split("-")[1])
    out = (np.full(6 * P, 7.0), np.zeros(6 * P), np.zeros(1))
    with pytest.raises(gpu.GPError):
        free.step(rec_f_dev, out=out)
    _, bh, ch = _host_system(rec_f, free_pairs, P)
    assert np.abs(out[1] - bh).max() <= 1e-12 * np.abs(bh).max() and abs(out[2][0] - ch) <= 1e-12 * abs(ch) and np.all(out[0] == 7.0)
    x, _, _ = free.step(rec_f_dev, lam=1.0)  # ... and lambda cures it: (0 + I) x = b
    assert np.abs(x - bh).max() <= 1e-12 * np.abs(bh).max()


def _c3_like_slots(n):
    pairs = [(i, j) for i in range(n) for j in range(i + 1, min(i + 5, n))]
    pairs = (pairs + [(j, i) for i, j in pairs])[: 4 * n]
    return [(a - 1, b - 1) for a, b in pairs]  # pose 0 held


@pytest.mark.parametrize("ordering", ["auto", "natural", "nd", "amd", "amd1"])
@pytest.mark.parametrize("graph", ["c3:64", "c1:2", "band:16", "band:100", "chain:128", "freechain:40", "isolated"])
def test_one_launch_step_is_bit_identical(gpu, graph, ordering):
    """a system whose factor fits the LDS of one compute unit runs its damped step as ONE launch (sparse_small_step_kernel: assembly, damping, the levels of the
    block-sparse LL^T, both substitutions, x / b / c / status to the host).  Same operations in the same order as the multi-launch form: x, b, c bit for bit, with every
    damping form and a prior, for BASELINE configs[2]'s graph (64 poses, 256 factors), configs[0]'s (one free pose) and others; against numpy as well."""
    import torch

    rng = np.random.default_rng(17)
    kind, _, num = graph.partition(":")
    if kind == "c3" or kind == "c1":
        n = int(num)
        slots, P = _c3_like_slots(n), n - 1
    elif kind == "band":
        P = int(num)
        slots = [(-1, 0)] + [(i, i + d) for i in range(P) for d in (1, 2, 5) if i + d < P]
    elif kind == "chain":
        P = int(num)
        slots = [(-1, 0)] + [(i, i + 1) for i in range(P - 1)]
    elif kind == "freechain":  # no pose held: more than eight one-column subtrees under the minimum-degree orderings (several batches of lists per level)
        P = int(num)
        slots = [(i, i + 1) for i in range(P - 1)]
    else:
        P = 12
        slots = [(-1, 0), (0, 1), (1, 2), (-1, 5), (5, 6), (6, 7), (7, 5), (-1, 11), (-1, 3), (3, 4), (-1, 8), (8, 9), (9, 10)]
    rec = _random_records(slots, rng)
    rec_dev = torch.from_numpy(rec).cuda()
    one, multi = gpu.SparseLinearSystemGPU(P, slots, ordering=ordering), gpu.SparseLinearSystemGPU(P, slots, ordering=ordering)
    if not one.set_one_launch(True):
        sym = gpu.solver.sparse_symbolic(P, slots, gpu.SparseLinearSystemGPU.ORDERINGS[ordering])
        assert sym["nnz_l_blocks"] > 340, sym  # (only a factor too large for the LDS may decline: the dissection of the 64-pose band graph, 522 blocks; a 128-pose chain's 369)
        pytest.skip(f"{sym['nnz_l_blocks']} blocks of L do not fit one compute unit's LDS: multi-launch only")
    assert multi.set_one_launch(False) is False
    teams = gpu.SparseLinearSystemGPU(P, slots, ordering=ordering)  # the one-launch step's first form: every list a team of waves in lock step
    assert teams.set_one_launch("teams") is True
    lone = gpu.SparseLinearSystemGPU(P, slots, ordering=ordering)  # ... its second: a work list per lone wave (the default form teams waves up where a level has <= 4 lists)
    assert lone.set_one_launch("lone-waves") is True
    prior = rng.uniform(0.0, 2.0, 6 * P)
    Ah, bh, ch = _host_system(rec, slots, P)
    for lam, diag, pr in [(1e-5, False, None), (0.0, False, None), (1e-3, False, None), (10.0, True, None), (1e-2, False, prior), (0.5, True, prior)]:
        x1, b1, c1 = one.step(rec_dev, lam=lam, diagonal_damping=diag, prior_diag=pr)
        xm, bm, cm = multi.step(rec_dev, lam=lam, diagonal_damping=diag, prior_diag=pr)
        assert np.array_equal(x1, xm) and np.array_equal(b1, bm) and c1 == cm, (lam, diag, pr is not None, float(np.abs(x1 - xm).max()))
        assert np.array_equal(teams.step(rec_dev, lam=lam, diagonal_damping=diag, prior_diag=pr)[0], xm)
        assert np.array_equal(lone.step(rec_dev, lam=lam, diagonal_damping=diag, prior_diag=pr)[0], xm)
        damp = lam * np.diag(np.clip(np.diag(Ah), 1e-6, 1e32)) if diag else lam * np.eye(6 * P)
        if kind != "freechain" or lam > 0.0 or pr is not None:  # (the free chain's undamped system has the gauge freedom: both forms factor it alike, numpy has no say)
            want = np.linalg.solve(Ah + damp + (np.diag(pr) if pr is not None else 0.0), bh)
            assert np.linalg.norm(x1 - want) <= 1e-7 * np.linalg.norm(want)
        assert np.abs(b1 - bh).max() <= 1e-12 * np.abs(bh).max() and abs(c1 - ch) <= 1e-12 * abs(ch)
    # twice the same: nothing is left in the system between steps
    again = one.step(rec_dev, lam=1e-5)[0]
    assert np.array_equal(again, multi.step(rec_dev, lam=1e-5)[0])
    # an indeterminate system is reported by both forms alike, b and c arrive all the same
    rec0 = rec.copy()
    rec0[:, 2:110] = 0.0
    rec0_dev = torch.from_numpy(rec0).cuda()
    for sysm in (one, multi, teams, lone):
        out = (np.full(6 * P, 7.0), np.zeros(6 * P), np.zeros(1))
        with pytest.raises(gpu.GPError):
            sysm.step(rec0_dev, out=out)
        assert np.all(out[0] == 7.0) and np.abs(out[1] - bh).max() <= 1e-12 * np.abs(bh).max()
    assert np.array_equal(one.step(rec0_dev, lam=1.0)[0], multi.step(rec0_dev, lam=1.0)[0])


def test_one_launch_step_declines_a_factor_that_does_not_fit(gpu):
    P = 512
    slots = [(-1, 0)] + [(i, i + d) for i in range(P) for d in (1, 2, 7) if i + d < P]
    sp = gpu.SparseLinearSystemGPU(P, slots)
    assert sp.set_one_launch(True) is False  # 512 poses: the multi-launch schedule stays in charge


def test_one_launch_step_fuzz_against_the_multi_launch_form(gpu):
    """randomised soak of the LDS-resident step: 60 random pose graphs (a spanning tree + random loop closures + unary factors, 1 .. 110 poses, slots shuffled, some poses held
    fixed), every ordering in turn: where the factor fits, x / b / c are the multi-launch form's bit for bit, and x solves the host-assembled system"""
    import torch

    rng = np.random.default_rng(2026)
    orderings = ["auto", "natural", "nd", "amd", "amd1"]
    ran = 0
    for trial in range(60):
        P = int(rng.integers(1, 111))
        perm = rng.permutation(P)
        slots = [(-1, int(perm[0]))]  # one pose tied to a fixed one: the system is regular
        for k in range(1, P):
            slots.append((int(perm[rng.integers(0, k)]), int(perm[k])))  # spanning tree
        for _ in range(int(rng.integers(0, P // 2 + 2))):
            a, b = rng.integers(0, P, 2)
            if a != b:
                slots.append((int(a), int(b)))
        for _ in range(int(rng.integers(0, 3))):
            slots.append((-1, int(rng.integers(0, P))))
        rec = _random_records(slots, rng)
        rec_dev = torch.from_numpy(rec).cuda()
        o = orderings[trial % len(orderings)]
        one, multi = gpu.SparseLinearSystemGPU(P, slots, ordering=o), gpu.SparseLinearSystemGPU(P, slots, ordering=o)
        if not one.set_one_launch(True):
            continue
        multi.set_one_launch(False)
        lam = float(10.0 ** rng.uniform(-6, 0))
        diag = bool(trial % 3 == 0)
        x1, b1, c1 = one.step(rec_dev, lam=lam, diagonal_damping=diag)
        xm, bm, cm = multi.step(rec_dev, lam=lam, diagonal_damping=diag)
        assert np.array_equal(x1, xm) and np.array_equal(b1, bm) and c1 == cm, (trial, P, o, len(slots))
        Ah, bh, _ = _host_system(rec, slots, P)
        damp = lam * np.diag(np.clip(np.diag(Ah), 1e-6, 1e32)) if diag else lam * np.eye(6 * P)
        want = np.linalg.solve(Ah + damp, bh)
        assert np.linalg.norm(x1 - want) <= 1e-8 * np.linalg.norm(want), (trial, P, o)
        ran += 1
    assert ran >= 40  # (most random graphs of this size fit one compute unit's LDS)


@pytest.mark.parametrize("num_factors", [1, 3, 300, 777])
def test_one_pose_dense_step_is_bit_identical(gpu, num_factors):
    """a system of ONE pose (BASELINE configs[0] as an optimisation: a scan onto a map) runs its damped step as ONE launch (dense_one_pose_step_kernel): x, b, c bit for bit
    what the nine stream operations of the multi-launch path give -- unary factors on either side, every damping form, an indeterminate system -- and numpy's solve"""
    import torch

    rng = np.random.default_rng(5 + num_factors)
    slots = [((-1, 0) if rng.random() < 0.6 else (0, -1)) for _ in range(num_factors)]
    rec = _random_records(slots, rng, rows=12)
    rec_dev = torch.from_numpy(rec).cuda()
    one, multi = gpu.DenseLinearSystemGPU(1, slots), gpu.DenseLinearSystemGPU(1, slots)
    assert one.set_one_launch(True) is True and multi.set_one_launch(False) is False
    Ah, bh, ch = _host_system(rec, slots, 1)
    for lam, diag in [(1e-5, False), (0.0, False), (1e-2, False), (3.0, True)]:
        x1, b1, c1 = one.step(rec_dev, lam=lam, diagonal_damping=diag)
        xm, bm, cm = multi.step(rec_dev, lam=lam, diagonal_damping=diag)
        assert np.array_equal(x1, xm) and np.array_equal(b1, bm) and c1 == cm, (lam, diag, float(np.abs(x1 - xm).max()))
        damp = lam * np.diag(np.clip(np.diag(Ah), 1e-6, 1e32)) if diag else lam * np.eye(6)
        assert np.linalg.norm(x1 - np.linalg.solve(Ah + damp, bh)) <= 1e-9 * np.linalg.norm(x1)
        assert np.abs(b1 - bh).max() <= 1e-12 * np.abs(bh).max() and abs(c1 - ch) <= 1e-12 * abs(ch)
    # the three calls after a one-launch step still see a consistent system (A, b, c are left where build() leaves them)
    x3 = one.build(rec_dev, lam=1e-3).solve()
    assert np.array_equal(x3, multi.step(rec_dev, lam=1e-3)[0])
    # a prior takes the multi-launch path in both
    prior = rng.uniform(0.0, 2.0, 6)
    assert np.array_equal(one.step(rec_dev, lam=1e-3, prior_diag=prior)[0], multi.step(rec_dev, lam=1e-3, prior_diag=prior)[0])
    rec0 = rec.copy()
    rec0[:, 2:110] = 0.0
    rec0_dev = torch.from_numpy(rec0).cuda()
    for sysm in (one, multi):
        out = (np.full(6, 7.0), np.zeros(6), np.zeros(1))
        with pytest.raises(gpu.GPError):
            sysm.step(rec0_dev, out=out)
        assert np.all(out[0] == 7.0) and np.abs(out[1] - bh).max() <= 1e-12 * np.abs(bh).max()
    assert np.array_equal(one.step(rec0_dev, lam=1.0)[0], multi.step(rec0_dev, lam=1.0)[0])
