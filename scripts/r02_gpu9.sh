#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_knn_gicp_gpu.py tests/test_voxelmap_gpu.py tests/test_configs_gpu.py tests/test_cloud_gpu.py -q -m gpu -x --durations=5 > gpurun_out/r02_pytest9.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest9.log
tail -25 gpurun_out/r02_pytest9.log
cd /tmp
for w in map c5; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof9_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof9_$w -o $w -- python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py $w 10 > $GRAFT_REPO_ROOT/gpurun_out/r02_prof9_$w.log 2>&1
  grep "voxel map build\|C5 1M" $GRAFT_REPO_ROOT/gpurun_out/r02_prof9_$w.log
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof9_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -9 "$f" | cut -c1-150
done
python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py map 20; python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py c5 10
