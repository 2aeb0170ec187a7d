#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
for mode in 0 1 2; do
  rm -rf /tmp/pk
  GP_KNN_MODE=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py c5 5 > /tmp/pk.log 2>&1
  grep "C5 1M" /tmp/pk.log
  f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
  python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'tiled' in r['Name'] or 'covariance_kernel' in r['Name'] or 'gicp' in r['Name']: print('mode $mode', r['Name'][:40], 'calls', r['Calls'], 'avg_us', float(r['AverageNs'])/1e3, 'min_us', float(r['MinNs'])/1e3)
"
done
