#!/bin/bash
# quick GPU check of the structure builds: tests that cover binning / sort / map / k-NN, then the build timings with per-kernel stats
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/build_check; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_voxelmap_gpu.py tests/test_knn_gicp_gpu.py tests/test_cloud_gpu.py tests/test_vgicp_gpu.py tests/test_mirror_gpu.py -x -q 2>&1 | tail -5
timeout 200 python scripts/r04_map_build.py 2>/dev/null | grep "^{"
rm -rf /tmp/pm && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o mb -- python scripts/r04_map_build.py > $O/map_build.log 2>&1
f=$(find /tmp/pm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/map_build_kernel_stats.csv && python - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
t=$(find /tmp/pm -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python scripts/r04_build_timeline.py $t | tee $O/build_timeline.txt
