#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for p in 1 2 4 8 16; do
  echo "parts $p"
  GP_FINALIZE_PARTS=$p python scripts/trace_finalize.py 2>&1 | grep "total" | tail -1
  GP_FINALIZE_PARTS=$p timeout 600 python bench.py --no-c4 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('  step', d['ms_per_step'], 'pass', d['roofline']['device_pass_ms'], 'fin', d['roofline']['finalize_kernel_ms'], 'tile', d['roofline']['kernel_ms'])"
done
