#!/bin/bash
# C5: knockouts of the row-tiled covariance kernel + batched loads in the per-lane searches
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest -m gpu -x -q tests/test_knn_gicp_gpu.py 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
cd /tmp
for cfg in "0 0" "3 0"; do
  set -- $cfg
  rm -rf /tmp/pk
  GP_KNN_MODE=$1 GP_KNN_KNOCK=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o c5 -- python $GRAFT_REPO_ROOT/scripts/r02_profile_aux.py c5 5 > /tmp/pk.log 2>&1
  grep "C5 1M" /tmp/pk.log | sed "s/^/mode $1 knock $2: /"
  f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'covariance' in r['Name'] or 'gicp' in r['Name']: print('   ', r['Name'][:40], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'min_us', round(float(r['MinNs'])/1e3,1))
PY
done
