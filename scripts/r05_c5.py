"""Round 5, C5: gp_estimate_covariances (k = 10) with a given build of the library (--lib <file under gtsam_points_amd/>): wall per call on the config's cloud (the 1 M-point C2
source), on the denser, map-like target sampling of the same scene (configs.C5.covariances.ms_target_cloud) and on a real kitti_00 scan, and a SHA-256 of the covariance
arrays so that two builds can be held against each other bit for bit (rocprofv3 --stats of this script gives the kernel's own duration).  One JSON object per line."""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gtsam_points_amd import _capi
lib_name = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else "libgtsam_points_hip.so"
_capi.LIB_PATH = os.path.join(ROOT, "gtsam_points_amd", lib_name)
import gtsam_points_amd as gpa
from gtsam_points_amd import synthetic
d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
kitti = os.path.join(ROOT, "tests", "golden", "kitti_00", "000000.bin")
clouds = [("c5_source", d["source_points"]), ("c5_target", d["target_points"])]
if os.path.exists(kitti):
    clouds.append(("kitti_00", np.fromfile(kitti, dtype=np.float32).reshape(-1, 3)))
frames = [(name, gpa.PointCloudGPU(p)) for name, p in clouds]
for name, fr in frames:
    for _ in range(3):
        gpa.estimate_covariances_gpu(fr, 10)
ts = {name: [] for name, _ in frames}
for rep in range(9):
    for name, fr in frames:  # alternating, as bench.py does: neither call finds the other's scratch arrays waiting
        torch.cuda.synchronize()
        t = time.perf_counter()
        short = gpa.estimate_covariances_gpu(fr, 10)
        ts[name].append(time.perf_counter() - t)
for name, fr in frames:
    cov = fr.download("covs")
    print(json.dumps(dict(lib=lib_name, cloud=name, points=int(fr.size()), ms_median=round(float(np.median(ts[name])) * 1e3, 4), ms_min=round(float(np.min(ts[name])) * 1e3, 4),
                          sha256=hashlib.sha256(np.ascontiguousarray(cov).tobytes()).hexdigest()[:16])), flush=True)
