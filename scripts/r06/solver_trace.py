"""Round 6: where the one-launch step's time goes -- thread 0's shader-clock stamps (gp_debug_sparse_step_trace) on the C3 graph's structure."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gtsam_points_amd as gpa  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from solver_step_time import graph, records  # noqa: E402  (prints its own lines first when imported: harmless)

lib = gpa.load()
for n in (64, 16):
    slots = graph(n)
    rec_dev = torch.from_numpy(records(slots)).cuda()
    sp = gpa.SparseLinearSystemGPU(n - 1, slots)
    FORM = sys.argv[1] if len(sys.argv) > 1 else True  # "teams" / "wave-teams": the other forms
    assert sp.set_one_launch(FORM)
    tr = torch.zeros(64, dtype=torch.int64, device="cuda")
    out = (np.zeros(sp.size), np.zeros(sp.size), np.zeros(1))
    for _ in range(50):
        sp.step(rec_dev, lam=1e-5, out=out)
    lib.gp_debug_sparse_step_trace(sp._h, C.c_void_p(tr.data_ptr()))
    sp.step(rec_dev, lam=1e-5, out=out)
    torch.cuda.synchronize()
    w = tr.cpu().numpy().astype(np.int64)
    lib.gp_debug_sparse_step_trace(sp._h, None)
    ph = [int(w[i] - w[0]) for i in range(7)]
    rounds = [[int(w[8 + 4 * r + q] - w[8 + 4 * r]) for q in range(1, 4)] + [int(w[8 + 4 * (r + 1)] - w[8 + 4 * r]) if r < 13 and w[8 + 4 * (r + 1)] else None] for r in range(14) if w[8 + 4 * r]]
    print(json.dumps(dict(form=str(FORM), poses=n, phase_clocks=dict(system_in_lds=ph[5], factored=ph[2], substituted=ph[3], end=ph[4]), first_level_rounds_clocks_gather_diag_below_next=rounds)))
