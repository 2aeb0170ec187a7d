# Round 5, C5: the far kernel's bound tightened after every list (a lane holding k candidates bounds the group's k-th distance) against the build before it
# (libgtsam_points_hip_prev.so): wall per call, bit-identity of the covariances (scripts/r05_c5.py), the far queries' lives (scripts/r05_c5_farlog.py), the kernels' spans.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05n; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  timeout 200 python scripts/r05_c5.py --lib libgtsam_points_hip_prev.so 2>/dev/null | grep '^{' >> $O/ab.jsonl
  timeout 200 python scripts/r05_c5.py 2>/dev/null | grep '^{' >> $O/ab.jsonl
done
for c in c5_source c5_target kitti_00; do timeout 200 python scripts/r05_c5_farlog.py $c 2>/dev/null | grep -v amdgpu >> $O/farlog.txt; done
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o s -- python scripts/r05_c5.py > $O/rocprof.log 2>&1
python scripts/r05_c5_timeline.py $(ls /tmp/pk/*kernel_trace.csv /tmp/pk/*/*kernel_trace.csv 2>/dev/null | head -1) > $O/timeline.txt 2>&1
timeout 600 python -m pytest tests/test_knn_gicp_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest_knn.txt 2>&1; echo "pytest rc $?" >> $O/pytest_knn.txt
cat $O/ab.jsonl; cat $O/timeline.txt; grep -v "^long" $O/farlog.txt | cut -c1-400; tail -3 $O/pytest_knn.txt
