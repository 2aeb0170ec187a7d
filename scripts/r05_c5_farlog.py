"""Round 5, C5: per-query stamps of covariance_far_kernel (measurement build libgtsam_points_hip_wavelog.so): phase A (block shells) | phase B (superblock shells) | merge,
candidates scanned by the group's first lane, superblock shells walked.  Usage: python scripts/r05_c5_farlog.py [c5_source|c5_target|kitti_00]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gtsam_points_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "gtsam_points_amd", "libgtsam_points_hip_wavelog.so")
import gtsam_points_amd as gpa
from gtsam_points_amd import synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "c5_source"
if which == "kitti_00":
    pts = np.fromfile(os.path.join(ROOT, "tests", "golden", "kitti_00", "000000.bin"), dtype=np.float32).reshape(-1, 3)
else:
    d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
    pts = d["source_points"] if which == "c5_source" else d["target_points"]
src = gpa.PointCloudGPU(pts)
n = src.size()
W = (n + 63) // 64 + 2
for _ in range(3):
    gpa.estimate_covariances_gpu(src, 10)
buf = torch.zeros(8 + 8 * (W + n), dtype=torch.int64, device="cuda")
gpa.estimate_covariances_gpu(src, 10, counters=buf)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
far = raw[8 + 8 * W:].reshape(-1, 8)
far = far[far[:, 0] > 0]
if len(far) == 0:
    print(json.dumps(dict(cloud=which, far_queries=0)))
    sys.exit(0)
t0 = far[:, 0].min()
a = (far[:, 1] - far[:, 0]) / 100.0
b = (far[:, 2] - far[:, 1]) / 100.0
m = (far[:, 3] - far[:, 2]) / 100.0
life = (far[:, 3] - far[:, 0]) / 100.0
print(json.dumps(dict(cloud=which, points=int(n), far_queries=int(len(far)), kernel_span_us=round(float((far[:, 3].max() - t0) / 100.0), 1), settled_in_phase_a=int(far[:, 6].sum()),
                      life_mean=round(float(life.mean()), 1), life_p50=round(float(np.median(life)), 1), life_p99=round(float(np.percentile(life, 99)), 1), life_max=round(float(life.max()), 1),
                      phase_a_mean=round(float(a.mean()), 1), phase_b_mean=round(float(b.mean()), 1), merge_mean=round(float(m.mean()), 1),
                      cands_lane0_mean=round(float(far[:, 4].mean()), 1), cands_lane0_max=int(far[:, 4].max()), b_shells_mean=round(float(far[:, 5].mean()), 2), b_shells_max=int(far[:, 5].max()))))
for j in np.argsort(-life)[:8]:
    print("long", dict(life=round(float(life[j]), 1), phase_a=round(float(a[j]), 1), phase_b=round(float(b[j]), 1), merge=round(float(m[j]), 1), cands_lane0=int(far[j, 4]), b_shells=int(far[j, 5]),
                       start=round(float((far[j, 0] - t0) / 100.0), 1)))
