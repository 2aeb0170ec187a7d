#!/bin/bash
# Round 6: the whole GPU suite, smoke, and the driver's bench command; outputs under gpurun_out/r06/
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r06; mkdir -p $O
( time timeout 2400 python3 -m pytest tests -q -m gpu --durations=15 ) > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt
tail -30 $O/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -2 $O/smoke.txt
