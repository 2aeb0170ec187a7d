#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
for w in map c5; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$w -o $w -- python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py $w 10 > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_$w.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/r02_prof_$w.log
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-200
done
