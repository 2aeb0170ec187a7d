"""Deterministic synthetic LiDAR workloads for the bench and the large-size tests (numpy only; not on the hot path).

"street box" scene of SURVEY.md section 8(d) C2: ground plane z = -1.7 m (+ N(0, 0.02) range noise), 24 vertical wall
rectangles, sensor at the origin.  Points are produced by casting a spinning multi-beam pattern (ring-major order,
like the raw KITTI scans under data/kitti_00: consecutive points sweep azimuth on one ring), so memory order has the
spatial coherence of a real scan.  Covariances come from the generating surface normal, C = I - 0.999 n n^T, which is
exactly what estimate_covariances' (1e-3, 1, 1) eigenvalue regularisation yields (covariance_estimation.cpp:49-53).
"""
import numpy as np

GROUND_Z = -1.7
MAX_RANGE = 80.0
MIN_RANGE = 3.0


def make_walls(seed=42, num_walls=24, origin=(0.0, 0.0)):
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0, 2 * np.pi, num_walls)
    dist = rng.uniform(8.0, 60.0, num_walls)
    centers = np.stack([origin[0] + dist * np.cos(ang), origin[1] + dist * np.sin(ang)], 1)
    heading = ang + rng.uniform(-0.6, 0.6, num_walls)  # wall normal roughly faces the sensor
    half_width = rng.uniform(4.0, 20.0, num_walls)
    height = rng.uniform(2.0, 12.0, num_walls)
    return dict(centers=centers, normals=np.stack([np.cos(heading), np.sin(heading)], 1), half_width=half_width, top=GROUND_Z + height)


def cast_scan(num_points, seed=42, sensor_pose=np.eye(4), rings=128, walls=None, noise=0.02, oversample=1.35):
    """Return (points float32 [N,3] in the SENSOR frame, normals float32 [N,3] in the sensor frame), ring-major order."""
    for _ in range(6):
        out = _cast_scan(num_points, seed, sensor_pose, rings, walls, noise, oversample)
        if out is not None:
            return out
        oversample *= 1.6
    raise RuntimeError("cast_scan: could not produce enough valid returns")


def _cast_scan(num_points, seed, sensor_pose, rings, walls, noise, oversample):
    walls = make_walls(42) if walls is None else walls
    rng = np.random.default_rng(seed)
    T = np.asarray(sensor_pose, dtype=np.float64)
    R, t = T[:3, :3], T[:3, 3]
    steps = int(np.ceil(num_points * oversample / rings))
    el = np.radians(np.linspace(-24.8, 4.0, rings))
    az = np.linspace(-np.pi, np.pi, steps, endpoint=False)
    pts_out, nrm_out = [], []
    # only walls that can be hit from this station
    along2 = np.stack([-walls["normals"][:, 1], walls["normals"][:, 0]], 1)
    rel = t[:2][None] - walls["centers"]
    a_cl = np.clip((rel * along2).sum(1), -walls["half_width"], walls["half_width"])
    near = np.linalg.norm(t[:2][None] - (walls["centers"] + a_cl[:, None] * along2), axis=1) <= MAX_RANGE
    wnorm, wcen = walls["normals"][near], walls["centers"][near]
    whalf, wtop = walls["half_width"][near], walls["top"][near]
    wn = np.concatenate([wnorm, np.zeros((len(wnorm), 1))], 1)  # (W,3)
    wc = np.concatenate([wcen, np.zeros((len(wcen), 1))], 1)
    walong = np.stack([-wnorm[:, 1], wnorm[:, 0], np.zeros(len(wn))], 1)
    plane_off = ((wc - t) * wn).sum(1)[None, :]       # n.(c - o)
    along_off = ((t - wc) * walong).sum(1)[None, :]   # (o - c).along
    for ring in range(rings):
        jitter = rng.uniform(-0.5, 0.5, steps) * (2 * np.pi / steps)
        a = az + jitter
        d_s = np.stack([np.cos(el[ring]) * np.cos(a), np.cos(el[ring]) * np.sin(a), np.full(steps, np.sin(el[ring]))], 1)
        d = d_s @ R.T  # world direction
        best_t = np.full(steps, np.inf)
        best_n = np.zeros((steps, 3))
        # ground
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (GROUND_Z - t[2]) / d[:, 2]
        ok = (d[:, 2] < -1e-6) & (tg > 0)
        best_t[ok] = tg[ok]
        best_n[ok] = np.array([0.0, 0.0, 1.0])
        # walls
        denom = d @ wn.T  # (S,W)
        with np.errstate(divide="ignore", invalid="ignore"):
            tw = plane_off / denom
        along = along_off + tw * (d @ walong.T)
        hz = t[2] + tw * d[:, 2:3]
        okw = (np.abs(denom) > 1e-9) & (tw > 0) & (np.abs(along) <= whalf[None]) & (hz >= GROUND_Z) & (hz <= wtop[None])
        tw = np.where(okw, tw, np.inf)
        wi = np.argmin(tw, 1)
        twm = tw[np.arange(steps), wi]
        closer = twm < best_t
        best_t[closer] = twm[closer]
        best_n[closer] = wn[wi[closer]]
        valid = np.isfinite(best_t) & (best_t >= MIN_RANGE) & (best_t <= MAX_RANGE)
        rn = best_t[valid] + rng.normal(0.0, noise, valid.sum())
        p_s = d_s[valid] * rn[:, None]  # sensor frame
        n_s = best_n[valid] @ R  # world normal -> sensor frame (R^T n)
        # orient normals toward the sensor
        flip = (n_s * p_s).sum(1) > 0
        n_s[flip] *= -1.0
        pts_out.append(p_s)
        nrm_out.append(n_s)
    pts = np.concatenate(pts_out)
    nrm = np.concatenate(nrm_out)
    if len(pts) < num_points:
        return None
    # thin uniformly (keeps ring-major order) to exactly num_points
    keep = np.floor(np.arange(num_points) * (len(pts) / num_points)).astype(np.int64)
    return pts[keep].astype(np.float32), nrm[keep].astype(np.float32)


def covs_from_normals(normals):
    """C = I - 0.999 n n^T as float32 (N,3,3), exactly symmetric."""
    n = np.asarray(normals, dtype=np.float64)
    n = n / np.linalg.norm(n, axis=1, keepdims=True)
    c = np.eye(3)[None] - 0.999 * n[:, :, None] * n[:, None, :]
    c = c.astype(np.float32)
    return np.ascontiguousarray(0.5 * (c + c.transpose(0, 2, 1)))


def expmap(xi):
    """GTSAM Pose3::Expmap, xi = [omega, v]."""
    xi = np.asarray(xi, dtype=np.float64)
    w, v = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-10:
        R, V = np.eye(3) + W, np.eye(3)
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th**2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


C1B_PERTURBATION = np.array([0.01, -0.02, 0.015, 0.10, -0.05, 0.03])


def make_pair(num_source, num_target, seed=42, source_offset=C1B_PERTURBATION):
    """C2-style pair: target scan from the origin, source scan from a sensor displaced by Expmap(source_offset).
    Returns dict(target_points, target_covs, source_points, source_covs, source_normals, T_true) where T_true maps the
    source frame into the target frame (so delta = T_true aligns them)."""
    walls = make_walls(seed)
    T_src = expmap(source_offset)
    tp, tn = cast_scan(num_target, seed=seed + 1, sensor_pose=np.eye(4), walls=walls, rings=128)
    sp, sn = cast_scan(num_source, seed=seed + 2, sensor_pose=T_src, walls=walls, rings=128)
    return dict(
        target_points=tp,
        target_covs=covs_from_normals(tn),
        source_points=sp,
        source_covs=covs_from_normals(sn),
        source_normals=sn,
        T_true=T_src,
    )


def make_submap(num_points, seed, walls=None, sensor_pose=np.eye(4), shuffle=True):
    """C3/C4-style submap: a scan re-ordered randomly (the kitti_07_dump submaps are merged/downsampled clouds
    with no scan order: 97 % of consecutive points change voxel)."""
    p, n = cast_scan(num_points, seed=seed, sensor_pose=sensor_pose, walls=walls, rings=64)
    if shuffle:
        perm = np.random.default_rng(seed + 7).permutation(len(p))
        p, n = p[perm], n[perm]
    return p, covs_from_normals(n), n


def merge_walls(wall_sets):
    return {k: np.concatenate([w[k] for w in wall_sets]) for k in wall_sets[0]}


def make_street(num_stations, spacing=40.0, seed=42, walls_per_station=24):
    """A street of `num_stations` sensor stations `spacing` metres apart along +x, each with its own 24 walls around it.
    Returns (walls, station poses)."""
    poses, wall_sets = [], []
    for k in range(num_stations):
        T = np.eye(4)
        T[0, 3] = spacing * k
        T[1, 3] = 2.0 * np.sin(0.7 * k)
        yaw = 0.05 * k
        T[:3, :3] = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        poses.append(T)
        wall_sets.append(make_walls(seed + 101 * k, walls_per_station, origin=(T[0, 3], T[1, 3])))
    walls = merge_walls(wall_sets)
    # drop walls that pass within 6 m of any station (a sensor hugging a wall returns almost nothing beyond MIN_RANGE)
    keep = np.ones(len(walls["half_width"]), dtype=bool)
    along = np.stack([-walls["normals"][:, 1], walls["normals"][:, 0]], 1)
    for T in poses:
        rel = T[:2, 3][None] - walls["centers"]
        a = np.clip((rel * along).sum(1), -walls["half_width"], walls["half_width"])
        closest = walls["centers"] + a[:, None] * along
        keep &= np.linalg.norm(T[:2, 3][None] - closest, axis=1) > 6.0
    walls = {k: v[keep] for k, v in walls.items()}
    return walls, poses


def make_merged_cloud(num_points, walls, station_poses, frame_pose, seed):
    """Merge one scan per station (scan order kept inside each scan) and express it in the frame `frame_pose`
    (world_T_frame).  Returns points, covs, normals as float32."""
    per = int(np.ceil(num_points / len(station_poses)))
    F_inv = np.linalg.inv(np.asarray(frame_pose, dtype=np.float64))
    pts, nrm = [], []
    for k, T in enumerate(station_poses):
        n_k = min(per, num_points - per * k)
        if n_k <= 0:
            break
        p, n = cast_scan(n_k, seed=seed + 13 * k, sensor_pose=T, walls=walls, rings=128)
        M = F_inv @ T  # frame_T_sensor
        pts.append(p.astype(np.float64) @ M[:3, :3].T + M[:3, 3])
        nrm.append(n.astype(np.float64) @ M[:3, :3].T)
    p = np.concatenate(pts).astype(np.float32)
    n = np.concatenate(nrm).astype(np.float32)
    return p, covs_from_normals(n), n


def make_c2_workload(num_source=1_000_000, num_target=2_000_000, seed=42, num_stations=16):
    """BASELINE.json configs[1]: 1 M-point source vs a 2 M-point GaussianVoxelMap.
    Target = 16 merged scans along a street (world frame); source = scans from every second station, expressed in a
    frame displaced from the world frame by Expmap(C1B_PERTURBATION) -- so delta = T_true aligns them."""
    walls, poses = make_street(num_stations, seed=seed)
    tp, tc, _ = make_merged_cloud(num_target, walls, poses, np.eye(4), seed + 1)
    T_true = expmap(C1B_PERTURBATION)
    sp, sc, sn = make_merged_cloud(num_source, walls, poses[::2], T_true, seed + 2)
    return dict(target_points=tp, target_covs=tc, source_points=sp, source_covs=sc, source_normals=sn, T_true=T_true)


def make_c3_graph(num_submaps=64, num_factors=256, seed=43):
    """BASELINE.json configs[2]: a 256-factor submap graph.  64 synthetic submaps (~20-25 k points, scan order shuffled like the
    kitti_07_dump submaps) along a street, 6 m apart; factors (target i, source j) for j = i+1..i+4 and their reverses, cut to
    `num_factors`; poses = ground truth o Expmap(U(-0.02, 0.02)^6) (rng 8191, like the reference tests).
    Returns dict(clouds=[(points, covs)], pairs=[(i, j)], deltas=[4x4], stations=[4x4])."""
    rng = np.random.default_rng(8191)
    walls, stations = make_street(num_submaps, spacing=6.0, seed=seed)
    clouds = []
    for i, T in enumerate(stations):
        p, c, _ = make_submap(20000 + 80 * i, seed=1000 + i, walls=walls, sensor_pose=T)
        clouds.append((p, c))
    pairs = [(i, j) for i in range(num_submaps) for j in range(i + 1, min(i + 5, num_submaps))]
    pairs = (pairs + [(j, i) for i, j in pairs])[:num_factors]
    deltas = [np.linalg.inv(stations[i]) @ stations[j] @ expmap(rng.uniform(-0.02, 0.02, 6)) for i, j in pairs]
    return dict(clouds=clouds, pairs=pairs, deltas=deltas, stations=stations, walls=walls)


C4_SUBMAPS, C4_POINTS, C4_OUT = 512, 32768, 8


def c4_factor_pairs(num_submaps=C4_SUBMAPS, out_degree=C4_OUT, base=64):
    """BASELINE.json configs[3]: 4096 pairwise factors = 8 outgoing factors per source submap.  Factor (target t, source s) for
    t = s + k, k = 1..8 (s - k where s + k would leave the street of `base` stations the submap belongs to, see
    make_c4_submaps), listed source-major so that a contiguous slice of the list is the shard of a contiguous range of source
    submaps."""
    pairs = []
    for s in range(num_submaps):
        for k in range(1, out_degree + 1):
            t = s + k if (s % base) + k < base else s - k
            pairs.append((t, s))
    return pairs


def make_c4_submaps(indices, base=64, seed=44):
    """The C4 submaps with the given indices (any subset of 0..511): 32768 points each, 1.0 m voxels downstream.
    Casting 512 scans on the host takes minutes, so the 512 submaps are 8 generations of `base` cast scans along a street
    (stations 6 m apart); generation g > 0 re-uses the geometry of generation 0 with an independent 1 cm jitter per point and
    is shifted 10 km along y, so that all 512 clouds and maps are distinct memory with distinct contents and no two
    generations overlap.  Deterministic in (index, seed).  Returns {index: (points, covs, station pose 4x4)}."""
    walls, stations = make_street(base, spacing=6.0, seed=seed - 1)
    cache, out = {}, {}
    for idx in indices:
        g, b = divmod(int(idx), base)
        if b not in cache:
            p, c, _ = make_submap(C4_POINTS, seed=2000 + b, walls=walls, sensor_pose=stations[b])
            cache[b] = (p, c)
        p, c = cache[b]
        if g > 0:
            p = (p.astype(np.float64) + np.random.default_rng(3000 + int(idx)).normal(0, 0.01, p.shape)).astype(np.float32)
        T = stations[b].copy()
        T[1, 3] += 10000.0 * g  # the station pose of this generation (points are in the sensor frame: unaffected)
        out[int(idx)] = (p, c, T)
    return out


def c4_delta(submaps, t, s, rng_seed=8191):
    """relative pose of factor (target t, source s): ground truth o a small deterministic perturbation"""
    rng = np.random.default_rng(rng_seed + 4096 * t + s)
    return np.linalg.inv(submaps[t][2]) @ submaps[s][2] @ expmap(rng.uniform(-0.02, 0.02, 6))
