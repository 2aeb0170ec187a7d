"""Time of the on-device normal-equation build + solve for pose graphs of growing size: the dense blocked LL^T (gp_solver.hip) and
the block-sparse LL^T over the pose graph (gp_sparse.hip, natural order and nested dissection).
Graphs: `chain` = odometry chain with one fixed pose; `loops` = chain + i -> i+2 and i -> i+7 edges (the round-1 graph)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa  # noqa: E402

rng = np.random.default_rng(3)


def records(pairs):
    rec = np.zeros((len(pairs), 122))
    for k in range(len(pairs)):
        J = rng.normal(size=(24, 12))
        H = J.T @ J
        rec[k, 2:38], rec[k, 38:74], rec[k, 74:110] = H[:6, :6].T.reshape(36), H[6:, 6:].T.reshape(36), H[:6, 6:].T.reshape(36)
        rec[k, 110:122] = rng.normal(size=12)
    return rec


def time_system(sys_, rec_dev, reps=5):
    x = sys_.build(rec_dev, lam=1e-2).solve()
    torch.cuda.synchronize()
    tb, ts = [], []
    for _ in range(reps):
        t = time.perf_counter()
        sys_.build(rec_dev, lam=1e-2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        x = sys_.solve()
        t2 = time.perf_counter()
        tb.append(t1 - t)
        ts.append(t2 - t1)
    return x, float(np.median(tb)) * 1e3, float(np.median(ts)) * 1e3


for kind in ("chain", "loops"):
    for P in [64, 256, 512, 2048, 8192]:
        pairs = [(-1, 0)] + [(i, i + 1) for i in range(P - 1)]
        if kind == "loops":
            pairs += [(i, i + d) for i in range(P) for d in (2, 7) if i + d < P]
        rec_dev = torch.from_numpy(records(pairs)).cuda()
        row = dict(graph=kind, poses=P, factors=len(pairs))
        xd = None
        if P <= 512:
            xd, b, s = time_system(gpa.DenseLinearSystemGPU(P, pairs), rec_dev, 3)
            row.update(dense_build_ms=round(b, 3), dense_solve_ms=round(s, 3))
        for name in ("natural", "nd", "amd", "amd1", "auto"):
            sp = gpa.SparseLinearSystemGPU(P, pairs, ordering=name)
            x, b, s = time_system(sp, rec_dev)
            info = sp.info()
            row.update({f"sparse_{name}_build_ms": round(b, 3), f"sparse_{name}_solve_ms": round(s, 3), f"sparse_{name}_l_blocks": info["nnz_l_blocks"],
                        f"sparse_{name}_block_products": info["block_products"], f"sparse_{name}_subtrees": info["num_subtrees"], f"sparse_{name}_top_columns": info["top_columns"]})
            if xd is not None:
                row[f"sparse_{name}_vs_dense_rel"] = float(np.linalg.norm(x - xd) / np.linalg.norm(xd))
        print(json.dumps(row), flush=True)
