"""Round 6 soak of the device-resident LM trial (speculative linearise, polled waits, the step kernel's LDS handshakes in the loop): the library's loop on BASELINE
configs[2]'s graph N times from the same start -- every run must end on the bits of the first, with and without speculation."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench_lm  # noqa: E402
import gtsam_points_amd as gpa  # noqa: E402
from gtsam_points_amd import synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
g = synthetic.make_c3_graph()
clouds = [gpa.PointCloudGPU(p, c) for p, c in g["clouds"]]
maps = []
for c in clouds:
    m = gpa.GaussianVoxelMapGPU(1.0, target_points_drop_rate=0.0)
    m.insert(c)
    maps.append(m)
factors = [gpa.IntegratedVGICPFactorGPU(t, s, maps[t], clouds[s]) for t, s in g["pairs"]]
truth = np.stack(g["stations"][: len(clouds)])
v0 = truth @ bench_lm.expmap_many(np.random.default_rng(8191).uniform(-0.1, 0.1, (len(clouds), 6)))
v0[0] = truth[0]
lm = gpa.LevenbergMarquardtGraphGPU(factors, g["pairs"], len(clouds), fixed=(0,))
ref, s0 = lm.optimize(v0, max_iterations=30)
bad = 0
for i in range(N):
    lm.set_speculation(i % 3 != 0)
    v, s = lm.optimize(v0, max_iterations=30)
    if not np.array_equal(v, ref) or s != s0:
        bad += 1
print(json.dumps(dict(runs=N, iterations=s0["iterations"], mismatches=bad, final_error=s0["final_error"])))
