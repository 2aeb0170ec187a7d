#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python scripts/trace_finalize.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
timeout 1500 python -m pytest -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
timeout 600 python bench.py --no-c4 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline'])"
