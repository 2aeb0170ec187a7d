#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 4 8 4 8; do
GP_VARIANT=$v timeout 900 python scripts/bench_configs.py C3,C4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('variant $v', d['config'][:12], d['tile_kernel_ms'], d['ms_per_linearize'], d['roofline_frac'], d['parity_max_rel_err'])"
done
