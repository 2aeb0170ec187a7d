"""Round 6 soak of the one-launch step's forms (the LDS handshakes of the wave teams are new): N random pose graphs (the generator of
tests/test_solver_gpu.py::test_one_launch_step_fuzz_against_the_multi_launch_form), every ordering in turn, each form stepped several times against the multi-launch x."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gtsam_points_amd as gpa  # noqa: E402
from test_solver_gpu import _random_records  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(777)
orderings = ["auto", "natural", "nd", "amd", "amd1"]
ran = bad = 0
shapes = {}
for trial in range(N):
    if trial % 2:  # band graphs (bandwidth 1 .. 6, a few loop closures): few, long level-0 lists -- the wave teams' case, pipelined where the columns are narrow
        P = int(rng.integers(2, 80))
        bw = int(rng.integers(1, 7))
        slots = [(-1, 0)] + [(i, i + d) for i in range(P) for d in range(1, bw + 1) if i + d < P and rng.random() < 0.9]
        for _ in range(int(rng.integers(0, 3))):
            a, b = rng.integers(0, P, 2)
            if a != b:
                slots.append((int(a), int(b)))
        slots += [(i, i + 1) for i in range(P - 1)]  # (connected whatever was dropped)
    else:
        P = int(rng.integers(1, 111))
        perm = rng.permutation(P)
        slots = [(-1, int(perm[0]))]
        for k in range(1, P):
            slots.append((int(perm[rng.integers(0, k)]), int(perm[k])))
        for _ in range(int(rng.integers(0, P // 2 + 2))):
            a, b = rng.integers(0, P, 2)
            if a != b:
                slots.append((int(a), int(b)))
    rec_dev = torch.from_numpy(_random_records(slots, rng)).cuda()
    o = orderings[trial % len(orderings)]
    sp = gpa.SparseLinearSystemGPU(P, slots, ordering=o)
    if not sp.set_one_launch(True):
        continue
    lam = float(10.0 ** rng.uniform(-6, 0))
    sp.set_one_launch(False)
    xm = sp.step(rec_dev, lam=lam)[0].copy()
    for form in (True, "lone-waves", "teams"):
        sp.set_one_launch(form)
        for rep in range(4):
            x = sp.step(rec_dev, lam=lam)[0]
            if not np.array_equal(x, xm):
                bad += 1
                print(json.dumps(dict(mismatch=True, trial=trial, P=P, ordering=o, form=str(form), rep=rep, max_abs=float(np.abs(x - xm).max()))), flush=True)
    sym = gpa.solver.sparse_symbolic(P, slots, gpa.SparseLinearSystemGPU.ORDERINGS[o])
    key = min(sum(1 for w in sym["work_lists"] if w["level"] == 0), 9)
    shapes[key] = shapes.get(key, 0) + 1
    ran += 1
print(json.dumps(dict(graphs=ran, of=N, mismatches=bad, level0_lists_histogram=shapes)))
