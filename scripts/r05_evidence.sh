#!/bin/bash
# Round-5 evidence run on the GPU box (through gpurun).  Everything lands under gpurun_out/r05/ and is copied into profiles/r05_* afterwards (scripts/r05_collect.sh).
# (The round's A/B runs have their own files: profiles/r05_kernel_experiments.*, r05_arrival_probe.txt, r05_c5_*, r05_error_cache.txt, r05_rehearsal_n*.log.)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05; rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $O/device.txt; nproc >> $O/device.txt
B="--no-cpu-baseline --no-c4 --no-configs --no-traffic --no-cold"
# 1. rocprofv3 --stats of the driver's command (--steps 20 --warmup 5; fused step on the packed mirror), its per-dispatch trace, the same with --finalize two-kernel (the split of
#    the launch patterns) and with --device-warmup-ms 0 (rounds 1-3's protocol)
rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps 20 --warmup 5 $B > $O/rocprof_bench.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -3 $f | cut -c1-200
tf=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
rm -rf /tmp/prof2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o bench -- python bench.py --steps 20 --warmup 5 $B --finalize two-kernel > $O/rocprof_bench_two_kernel.log 2>&1
f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats_two_kernel.csv
t=$(find /tmp/prof2 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python scripts/kernel_trace_split.py "$t" $O/kernel_trace_split.json "$tf" | tee $O/kernel_trace_split.txt
rm -rf /tmp/prof4 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o bench -- python bench.py --steps 20 --warmup 5 $B --device-warmup-ms 0 > $O/rocprof_bench_no_warmup.log 2>&1
f=$(find /tmp/prof4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats_no_warmup.csv && head -2 $f | cut -c1-200
# 2. the driver's command itself, whole line (in-run PMC traffic, cold leg, configs incl. lm_c1 / lm_c3, c4, big source, CPU baseline), three times (box-to-box and run-to-run spread)
for i in 1 2 3; do ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_run$i.log 2>&1; grep "^{" $O/bench_run$i.log > $O/bench_run$i.json; done
cp $O/bench_run1.json $O/bench_n1.json; cut -c1-260 $O/bench_n1.json; grep -h "^real" $O/bench_run*.log
# 3. map build timeline (unchanged code this round: the figure the line carries)
rm -rf /tmp/pm && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o mb -- python scripts/r04_map_build.py > $O/map_build.log 2>&1
t=$(find /tmp/pm -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python scripts/r04_build_timeline.py $t > $O/build_timeline.txt
# 4. smoke and the GPU test-suite
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 1200 python -m pytest -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
ls $O
