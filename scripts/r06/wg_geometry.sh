#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06; mkdir -p $O
timeout 600 python3 scripts/r06/wg_geometry.py --reps 3 > $O/wg_geometry.jsonl 2> $O/wg_geometry.err; echo "exit $?"; cat $O/wg_geometry.jsonl | cut -c1-420
rm -rf /tmp/wgprof; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wgprof -o wg -- python3 $OLDPWD/scripts/r06/wg_geometry.py --reps 2 > $O/wg_geometry_rocprof.jsonl 2> $O/wg_geometry_rocprof.err)
f=$(find /tmp/wgprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/wg_geometry_kernel_stats.csv && grep stream_kernel $f | sed 's/(gp::FactorDesc.*)",/",/' | cut -c1-200
