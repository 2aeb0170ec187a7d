#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_voxelmap_gpu.py tests/test_vgicp_gpu.py tests/test_cloud_gpu.py tests/test_multi_gpu.py -q -m gpu -x > gpurun_out/r02_pytest8.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_pytest8.log
tail -8 gpurun_out/r02_pytest8.log
cd /tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_map3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_map3 -o map -- python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py map 20 > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_map3.log 2>&1
grep "voxel map build" $GRAFT_REPO_ROOT/gpurun_out/r02_prof_map3.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_map3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py map 20
