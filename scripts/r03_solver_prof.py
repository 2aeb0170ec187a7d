"""per-kernel times of one block-sparse solve (rocprofv3 --kernel-trace --stats around this): band graph i -> i+1, i+2, i+7, 512 poses"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
rng = np.random.default_rng(3)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ordering = sys.argv[2] if len(sys.argv) > 2 else "nd"
pairs = [(-1, 0)] + [(i, i + d) for i in range(P) for d in (1, 2, 7) if i + d < P]
rec = np.zeros((len(pairs), 122))
for k in range(len(pairs)):
    J = rng.normal(size=(24, 12)); H = J.T @ J
    rec[k, 2:38], rec[k, 38:74], rec[k, 74:110] = H[:6, :6].T.reshape(36), H[6:, 6:].T.reshape(36), H[:6, 6:].T.reshape(36)
    rec[k, 110:122] = rng.normal(size=12)
rec_dev = torch.from_numpy(rec).cuda()
sp = gpa.SparseLinearSystemGPU(P, pairs, ordering=ordering)
print(sp.info(), gpa.sparse_symbolic(P, pairs, gpa.SparseLinearSystemGPU.ORDERINGS[ordering])["critical_columns"])
for _ in range(20):
    sp.build(rec_dev, lam=1e-2).solve()
