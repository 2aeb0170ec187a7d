"""Round 4 / 5, C5: per-wave stage stamps of covariance_kernel (measurement build libgtsam_points_hip_wavelog.so: gp_knn.hip compiled with -DGP_KNN_WAVELOG).
Row per wave: start | fine shells done | block shells done | superblocks done | end | lanes left after the fine shells | after the block shells | first query."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gtsam_points_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "gtsam_points_amd", "libgtsam_points_hip_wavelog.so")
import gtsam_points_amd as gpa
from gtsam_points_amd import synthetic
st = int(sys.argv[1]) if len(sys.argv) > 1 else 0
d = synthetic.make_c2_workload(1_000_000, 1_000_000, seed=42)
src = gpa.PointCloudGPU(d["source_points"])
n = src.size()
W = (n + 63) // 64 + 2
for _ in range(3):
    gpa.estimate_covariances_gpu(src, 10, structure=st)
buf = torch.zeros(8 + 8 * W, dtype=torch.int64, device="cuda")
gpa.estimate_covariances_gpu(src, 10, structure=st, counters=buf)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
print("counters", raw[:8].tolist())
w = raw[8:].reshape(-1, 8)
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
start = (w[:, 0] - t0) / 100.0
end = (w[:, 4] - t0) / 100.0
life = end - start
fine = np.where(w[:, 1] > 0, (w[:, 1] - w[:, 0]) / 100.0, life)
sum32, sum64 = (w[:, 3] & 0xffffffff).astype(np.float64), (w[:, 3] >> 32).astype(np.float64)
max32, maxr = (w[:, 2] & 0xffffffff).astype(np.float64), (w[:, 2] >> 32).astype(np.float64)
print(json.dumps(dict(structure=st, waves=int(len(w)), kernel_us=round(float(end.max()), 1), life_mean=round(float(life.mean()), 1), life_p50=round(float(np.median(life)), 1),
                      life_p99=round(float(np.percentile(life, 99)), 1), life_max=round(float(life.max()), 1), sum_over_slots_us=round(float(life.sum() / 4096), 1),
                      fine_sum_over_slots=round(float(fine.sum() / 4096), 1), waves_past_fine=int((w[:, 5] > 0).sum()), lanes_past_fine=int(w[:, 5].sum()), 
                      f32_per_query=round(float(sum32.sum() / 1e6), 1), f64_per_query=round(float(sum64.sum() / 1e6), 1))))
span = (w[:, 7] >> 32).astype(np.float64); w[:, 7] &= 0xffffffff
pop0, blkpop0 = (w[:, 6] & 0xffffffff).astype(np.float64), (w[:, 6] >> 32).astype(np.float64)
print("life by cells spanned:")
for lo, hi in [(1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 63), (64, 64), (65, 10**9)]:
    m = (span >= lo) & (span <= hi)
    if m.any():
        print(f"  span {lo}-{hi}: waves {int(m.sum())}  life mean {life[m].mean():.0f} p90 {np.percentile(life[m], 90):.0f} max {life[m].max():.0f}  own-cell pop {pop0[m].mean():.0f}  own-block pop {blkpop0[m].mean():.0f}")
print("corr(life, sum32) %.3f  corr(life, max32) %.3f  corr(life, sum64) %.3f  corr(life, max shell) %.3f" % tuple(np.corrcoef(life, x)[0, 1] for x in (sum32, max32, sum64, maxr)))
# life by decile of max32
for lo, hi in [(0, 50), (50, 90), (90, 99), (99, 100)]:
    a_, b_ = np.percentile(life, lo), np.percentile(life, hi)
    m = (life >= a_) & (life <= b_)
    print(f"life p{lo}-p{hi}: {a_:.0f}-{b_:.0f} us  waves {int(m.sum())}  mean sum32/lane {sum32[m].mean() / 64:.0f}  mean max32 {max32[m].mean():.0f}  mean sum64/lane {sum64[m].mean() / 64:.1f}  mean last shell {maxr[m].mean():.2f}  share of all wave time {life[m].sum() / life.sum():.2f}")
order = np.argsort(-life)[:10]
for i in order:
    print(dict(first_query=int(w[i, 7]), start=round(float(start[i]), 1), life=round(float(life[i]), 1), sum32_per_lane=round(float(sum32[i] / 64)), max32=int(max32[i]), sum64_per_lane=round(float(sum64[i] / 64), 1),
               last_shell=int(maxr[i]), lanes_past_fine=int(w[i, 5])))
print("end percentiles", {p: round(float(np.percentile(end, p)), 1) for p in (50, 90, 99, 99.9, 100)})
late = np.argsort(-end)[:8]
for i in late:
    print("late", dict(first_query=int(w[i, 7]), start=round(float(start[i]), 1), end=round(float(end[i]), 1), life=round(float(life[i]), 1), sum32_per_lane=round(float(sum32[i] / 64)), last_shell=int(maxr[i])))

# round 5: where the longest waves spend their time (fine shells vs the coarser stages)
order = np.argsort(-life)[:12]
for j in order:
    print("long", dict(first_query=int(w[j, 7]), life=round(float(life[j]), 1), fine_part=round(float(fine[j]), 1), coarse_part=round(float(life[j] - fine[j]), 1),
                       sum32_per_lane=int(sum32[j] / 64), max32=int(max32[j]), last_shell=int(maxr[j]), lanes_past_fine=int(w[j, 5])))
heavy = life > 250
print("waves over 250 us:", int(heavy.sum()), "of them with lanes past the fine level:", int((heavy & (w[:, 5] > 0)).sum()), "mean fine part", round(float(fine[heavy].mean()), 1) if heavy.any() else None)
