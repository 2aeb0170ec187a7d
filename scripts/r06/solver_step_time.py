"""Round 6: what ONE damped step (gp_sparse_system_step) costs on the C3 graph's structure (64 poses, pose 0 held: 63 slots, 256 factors i -> i+1..i+4 and
back), on C1's (one free pose) and others, per ordering, as ONE launch (sparse_small_step_kernel, where the factor fits one compute unit's LDS) and in the multi-launch form;
run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations.  One JSON object per line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gtsam_points_amd as gpa  # noqa: E402

rng = np.random.default_rng(3)
ORDERINGS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["auto", "natural", "nd", "amd", "amd1"]


def records(pairs):
    rec = np.zeros((len(pairs), 122))
    for k in range(len(pairs)):
        J = rng.normal(size=(24, 12))
        H = J.T @ J
        rec[k, 2:38], rec[k, 38:74], rec[k, 74:110] = H[:6, :6].T.reshape(36), H[6:, 6:].T.reshape(36), H[:6, 6:].T.reshape(36)
        rec[k, 110:122] = rng.normal(size=12)
        rec[k, 1] = 1.0 + k
    return rec


def graph(n):
    pairs = [(i, j) for i in range(n) for j in range(i + 1, min(i + 5, n))]
    pairs = (pairs + [(j, i) for i, j in pairs])[: 4 * n]
    return [(a - 1, b - 1) for a, b in pairs]  # pose 0 held: slot -1


def timed(sp, rec_dev, out):
    for _ in range(20):
        sp.step(rec_dev, lam=1e-5, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t = time.perf_counter()
        sp.step(rec_dev, lam=1e-5, out=out)
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3, out[0].copy()


def main():
    for name, n in (("C3 (64 poses)", 64), ("C1 (2 poses)", 2), ("96 poses", 96), ("128 poses", 128), ("16 poses", 16)):
        slots = graph(n)
        rec = records(slots)
        rec_dev = torch.from_numpy(rec).cuda()
        for o in ORDERINGS:
            sp = gpa.SparseLinearSystemGPU(n - 1, slots, ordering=o)
            out = (np.zeros(sp.size), np.zeros(sp.size), np.zeros(1))
            row = dict(graph=name, ordering=o)
            x_one = None
            if sp.set_one_launch(True):
                med, mn, x_one = timed(sp, rec_dev, out)
                row.update(one_launch_ms=round(med, 4), one_launch_ms_min=round(mn, 4))
                if sp.set_one_launch("teams"):  # the one-launch step's first form: every list a team of waves in lock step
                    med, mn, x_teams = timed(sp, rec_dev, out)
                    row.update(one_launch_teams_ms=round(med, 4), teams_bit_identical=bool(np.array_equal(x_teams, x_one)))
                if sp.set_one_launch("lone-waves"):  # ... its second: a work list per (lone) wave
                    med, mn, x_lone = timed(sp, rec_dev, out)
                    row.update(one_launch_lone_waves_ms=round(med, 4), lone_waves_bit_identical=bool(np.array_equal(x_lone, x_one)))
            sp.set_one_launch(False)
            med, mn, x = timed(sp, rec_dev, out)
            row.update(multi_launch_ms=round(med, 4), multi_launch_ms_min=round(mn, 4), bit_identical=bool(np.array_equal(x, x_one)) if x_one is not None else None)
            A, b, c = sp.build(rec_dev, lam=1e-5).download()
            xr = np.linalg.solve(A, b)
            sym = gpa.solver.sparse_symbolic(n - 1, slots, gpa.SparseLinearSystemGPU.ORDERINGS[o])
            row.update(levels=sym["num_levels"], critical_columns=sym["critical_columns"], lists=sym["num_lists"], l_blocks=sym["nnz_l_blocks"], rel_err_vs_numpy=float(np.abs(x - xr).max() / np.abs(xr).max()))
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
