#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest -m gpu -x -q tests/test_vgicp_gpu.py -k "variant" 2>&1 | grep -E "passed|failed|Error|error|assert" | head
timeout 900 python scripts/r02_sweep.py 4,8,5,8,4 0 --big 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'trace' in d: print('trace', d['phase_median_us']); continue
    print(d['case'], d['variant'], d['tile_ms'], d['pass_ms'], d['sync_call_ms'], d['frac'], d.get('max_rel_err'))"
