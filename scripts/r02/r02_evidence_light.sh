#!/bin/bash
# Light refresh of the round-2 evidence after late changes (pytest, smoke, bench line, rocprofv3 kernel stats of the bench command, the
# batched configs): the subset of scripts/r02_evidence.sh whose numbers moved.  Lands under gpurun_out/r02/.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; grep -E "passed|failed" $O/pytest_gpu.txt | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.log 2>&1; grep "^{" $O/bench.log > $O/bench_n1.json; cut -c1-300 $O/bench_n1.json
rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-c4 > $O/rocprof_bench.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -4 $f | cut -c1-200
timeout 1500 python scripts/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cut -c1-300 $O/configs.jsonl
