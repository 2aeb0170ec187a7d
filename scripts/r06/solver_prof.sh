#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06; mkdir -p $O
timeout 600 python3 scripts/r06/solver_step_time.py > $O/solver_step_time${TAG:-}.jsonl 2> $O/solver_step_time.err; echo "exit $?"; cat $O/solver_step_time${TAG:-}.jsonl; tail -3 $O/solver_step_time.err
rm -rf /tmp/sprof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sprof -o s -- python3 $OLDPWD/scripts/r06/solver_step_time.py auto > /dev/null 2> $O/solver_prof.err)
f=$(find /tmp/sprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/solver_kernel_stats${TAG:-}.csv && sed 's/(gp::[^"]*"/"/; s/(double[^"]*"/"/' $f | cut -c1-150 | head -20
