"""Round 6: what one LM iteration of BASELINE configs[2]'s graph (256 factors / 64 poses, pose 0 held) costs host to host with
  host    the host-driven loop (bench_lm.GpuGraph, solver "device": poses up, linearise | step, wait | numpy retract, poses up, error evaluation, wait)
  trial   the values in device memory (gp_lm_graph_*: linearise | step + retract + error evaluation, ONE wait), the interpreter driving the three calls
  native  the library's own loop over the same three calls (gp_lm_graph_optimize)
best of five runs each, interleaved; prints one JSON object per line."""
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


def graph(name):
    if name == "c3":
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
        return factors, g["pairs"], len(clouds), truth, v0, (clouds, maps)
    raise SystemExit(name)


factors, pairs, n, truth, v0, keep = graph("c3")
gg = bench_lm.GpuGraph(gpa, factors, pairs, n, fixed=0, solver="device")
tg = bench_lm.GpuTrialGraph(gpa, factors, pairs, n, fixed=0)
runs = dict(host=lambda: bench_lm.run_lm(gg, v0, max_iterations=30), trial=lambda: bench_lm.run_lm(tg, v0, max_iterations=30), native=lambda: tg.native_loop(v0, max_iterations=30))
best = {}
for rep in range(6):
    for k, fn in runs.items():
        r = fn()
        if rep and (k not in best or r["seconds"] < best[k]["seconds"]):
            best[k] = r
for k, r in best.items():
    s = bench_lm.summarize(r, gg, truth, k)
    print(json.dumps(dict(loop=k, iterations=s["iterations"], inner=s["inner_iterations"], ms_per_iteration=s["ms_per_iteration"], phases=s["ms_per_iteration_by_phase"], gate_met=s["gate_met"],
                          rot=s["max_rotation_error_rad"], trans=s["max_translation_error_m"], final_error=r["final_error"])), flush=True)
