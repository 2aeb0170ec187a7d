#!/bin/bash
# copies the evidence run's files (gpurun_out/r05/, scratch) into profiles/r05_* (tracked) and refreshes the two files bench.py reads when it cannot measure itself
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r05
for f in bench_n1.json bench_run1.json bench_run2.json bench_run3.json bench_kernel_stats.csv bench_kernel_stats_two_kernel.csv bench_kernel_stats_no_warmup.csv kernel_trace_split.json kernel_trace_split.txt \
         build_timeline.txt smoke.txt pytest_gpu.txt device.txt; do
  [ -f $S/$f ] && cp $S/$f profiles/r05_$f
done
cp $S/kernel_trace_split.json profiles/kernel_trace_split.json
python - <<'PY'
import json
b = json.loads([l for l in open("profiles/r05_bench_n1.json") if l.startswith("{")][-1])
d = b["roofline"].get("traffic_detail")
if d:
    json.dump(dict(tile_kernel_hbm_bytes_per_launch=d["tile_kernel_hbm_bytes_per_launch"], fetch_size_kib=d["fetch_size_kib"], write_size_kib=d["write_size_kib"],
                   calibration_fetch_kib=d["calibration_fetch_kib"], fetch_scale=d["fetch_scale"], source="round-5 evidence run: bench.py's own in-run rocprofv3 --pmc passes (profiles/r05_bench_n1.json)"),
              open("profiles/hbm_traffic.json", "w"), indent=1)
PY
ls profiles/r05_* | wc -l
