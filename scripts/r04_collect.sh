#!/bin/bash
# copies the evidence run's files (gpurun_out/r04/, scratch) into profiles/r04_* (tracked) and refreshes the two files bench.py reads
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r04
for f in bench_n1.json bench_kernel_stats.csv bench_kernel_stats_two_kernel.csv bench_kernel_stats_no_mirror.csv bench_kernel_stats_no_warmup.csv kernel_trace_split.json kernel_trace_split.txt \
         hbm_traffic.json pmc_summary.txt pmc_tile_sq.txt sweep.jsonl sweep_8m.jsonl instep_xcd.jsonl instep_xcd_equal_shares.jsonl warm.jsonl dispatch_ramp_probe.txt \
         c5_staging.jsonl c5_wavelog.txt map_build.json build_timeline.txt sort_probe.txt bins_probe.txt map_build_kernel_stats.csv smoke.txt pytest_gpu.txt device.txt; do
  [ -f $S/$f ] && cp $S/$f profiles/r04_$f
done
cp $S/hbm_traffic.json profiles/hbm_traffic.json
cp $S/kernel_trace_split.json profiles/kernel_trace_split.json
ls profiles/r04_* | wc -l
