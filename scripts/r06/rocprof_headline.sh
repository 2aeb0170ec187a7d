#!/bin/bash
# Round 6: rocprofv3 --kernel-trace --stats of the HEADLINE protocol of the driver's command (bench.py --steps 20 --warmup 5 with the optional legs off: the 8 M-point leg
# runs the same kernel instantiation on another workload and would pollute the per-kernel average), and of the same with --device-warmup-ms 0 and --finalize two-kernel.
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06; mkdir -p $O; R=$PWD
run() {  # name, extra args
  rm -rf /tmp/hp; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -o h -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-c4 --no-traffic --no-rocprof --no-cpu-baseline --detail-file /tmp/d.json $2 > $O/$1.out 2> $O/$1.err)
  f=$(find /tmp/hp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/$1.csv && grep -E "Name|stream_kernel|finalize" $f | sed 's/(gp::FactorDesc[^"]*"/"/' | cut -c1-170
}
run bench_kernel_stats ""
run bench_kernel_stats_no_warmup "--device-warmup-ms 0"
run bench_kernel_stats_two_kernel "--finalize two-kernel"
