#!/bin/bash
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06; mkdir -p $O; R=$PWD
timeout 300 python3 scripts/r06/map_build_run.py 25 > $O/map_build_run.json 2> $O/map_build_run.err; cat $O/map_build_run.json
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/mbp; (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/mbp -o p -- python3 $R/scripts/r06/map_build_run.py 12 > /dev/null 2> $O/map_build_pmc_$ctr.err)
  f=$(find /tmp/mbp -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/map_build_pmc_$ctr.csv
done
rm -rf /tmp/mbs; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mbs -o s -- python3 $R/scripts/r06/map_build_run.py 25 > /dev/null 2> $O/map_build_stats.err)
f=$(find /tmp/mbs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/map_build_kernel_stats.csv
python3 scripts/r06/map_build_pmc_summary.py $O | tee $O/map_build_pmc.txt
