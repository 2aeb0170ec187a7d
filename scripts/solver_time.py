"""Time of the dense on-device normal-equation build + solve for chain-with-loops pose graphs of growing size."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gtsam_points_amd as gpa
rng = np.random.default_rng(3)
for P in [64, 256, 512]:
    pairs = [(i, i + d) for i in range(P) for d in (1, 2, 7) if i + d < P]
    rec = np.zeros((len(pairs), 122))
    for k in range(len(pairs)):
        J = rng.normal(size=(24, 12)); H = J.T @ J
        rec[k, 2:38], rec[k, 38:74], rec[k, 74:110] = H[:6, :6].T.reshape(36), H[6:, 6:].T.reshape(36), H[:6, 6:].T.reshape(36)
        rec[k, 110:122] = rng.normal(size=12)
    rec_dev = torch.from_numpy(rec).cuda()
    sys_ = gpa.DenseLinearSystemGPU(P, pairs)
    sys_.build(rec_dev, lam=1e-2).solve()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        sys_.build(rec_dev, lam=1e-2)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t) / 3
    t = time.perf_counter()
    for _ in range(3):
        x = sys_.build(rec_dev, lam=1e-2).solve()
    ts = (time.perf_counter() - t) / 3 - tb
    print(f"P = {P:4d} poses ({len(pairs)} factors, n = {6*P}): build {tb*1e3:.2f} ms, solve {ts*1e3:.2f} ms", flush=True)
