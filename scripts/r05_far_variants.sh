# Round 5, C5: covariance_far_kernel with more requests in flight per trip (v1: eight masks per lane, v2: eight candidates per lane, v3: both) against the product build;
# three alternating runs per build (scripts/r05_c5.py: wall per call, sha256 of the covariances).
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05q; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in libgtsam_points_hip.so libgtsam_points_hip_v1.so libgtsam_points_hip_v2.so libgtsam_points_hip_v3.so; do
  timeout 200 python scripts/r05_c5.py --lib $lib 2>/dev/null | grep '^{' >> $O/ab.jsonl
done; done
python - $O/ab.jsonl <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for cloud in ("c5_source", "c5_target", "kitti_00"):
    for lib in ("libgtsam_points_hip.so", "libgtsam_points_hip_v1.so", "libgtsam_points_hip_v2.so", "libgtsam_points_hip_v3.so"):
        v = [r["ms_median"] for r in rows if r["cloud"] == cloud and r["lib"] == lib]
        print(cloud, lib, min(v), max(v), set(r["sha256"] for r in rows if r["cloud"] == cloud))
PY
