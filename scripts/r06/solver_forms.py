"""Round 6: the one-launch step's forms side by side on BASELINE configs[2]'s structure: the product (a team of waves per list meeting without workgroup barriers; a lone wave per
list where a level has more than four), a lone wave per list throughout, teams of waves in lock step (the first form) and the multi-launch form; bit-identity of each against the multi-launch x."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gtsam_points_amd as gpa  # noqa: E402
from solver_step_time import graph, records, timed  # noqa: E402

for n in (64, 16, 33):
    slots = graph(n)
    rec_dev = torch.from_numpy(records(slots)).cuda()
    sp = gpa.SparseLinearSystemGPU(n - 1, slots)
    out = (np.zeros(sp.size), np.zeros(sp.size), np.zeros(1))
    sp.set_one_launch(False)
    ms, _, x_ref = timed(sp, rec_dev, out)
    row = dict(poses=n, multi_launch_ms=round(ms, 4))
    for form in (True, "lone-waves", "teams"):
        if sp.set_one_launch(form):
            ms, mn, x = timed(sp, rec_dev, out)
            row[str(form)] = dict(ms=round(ms, 4), ms_min=round(mn, 4), bit_identical=bool(np.array_equal(x, x_ref)))
    print(json.dumps(row), flush=True)
