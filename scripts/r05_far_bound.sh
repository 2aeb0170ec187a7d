# Round 5, C5: A/B of the product library against libgtsam_points_hip_prev.so (the build before the change under test): wall per call and bit-identity of the covariances
# (scripts/r05_c5.py, three alternating runs per build), the kernels' spans of one call (rocprofv3 --kernel-trace, scripts/r05_c5_timeline.py).  Output: gpurun_out/$1/
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05ab}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  timeout 200 python scripts/r05_c5.py --lib libgtsam_points_hip_prev.so 2>/dev/null | grep '^{' >> $O/ab.jsonl
  timeout 200 python scripts/r05_c5.py 2>/dev/null | grep '^{' >> $O/ab.jsonl
done
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o s -- python scripts/r05_c5.py > $O/rocprof.log 2>&1
python scripts/r05_c5_timeline.py $(ls /tmp/pk/*kernel_trace.csv /tmp/pk/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/timeline.txt 2>&1
python - $O/ab.jsonl <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for cloud in ("c5_source", "c5_target", "kitti_00"):
    for lib in ("libgtsam_points_hip_prev.so", "libgtsam_points_hip.so"):
        v = [r["ms_median"] for r in rows if r["cloud"] == cloud and r["lib"] == lib]
        print(cloud, lib, min(v), max(v), set(r["sha256"] for r in rows if r["cloud"] == cloud))
PY
cat $O/timeline.txt
