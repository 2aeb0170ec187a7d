# Round 5: configs.C5.covariances.ms inside bench.py read ~0.07 ms above scripts/r05_c5.py on the same box.  Bisected (gpurun_out/r05i/bisect.txt -> profiles/r05_c5_queue_pipes.txt):
# only with the C4 phase in front; traces with and without it (scripts/r05_trace_tail.py): after C4 the side stream sits on hardware queue 5 instead of 3 and the second covariance
# launch starts when the first one's grid is fully placed (100 / 160 us later), i.e. the two queues share a dispatch pipe.  This run: the same traces with the pipe probe in place.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05j; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_lm_gpu.py tests/test_knn_gicp_gpu.py -x -q -m gpu > $O/pytest_part.txt 2>&1; echo "pytest rc $?" >> $O/pytest_part.txt
T="--steps 5 --warmup 2 --no-cold --no-big-source --no-traffic"
timeout 400 python bench.py $T > $O/bench_c4_lm.log 2>&1
python - <<'PY' | tee $O/after.txt
import json
for l in open("gpurun_out/r05j/bench_c4_lm.log"):
    if l.startswith("{"):
        b = json.loads(l)
        c = b["configs"]["C5"]["covariances"]
        print("with C4 and LM in front: C5 covariances ms", c["ms"], c["ms_target_cloud"], c["ms_kitti_scan"], "side stream", c.get("side_stream"))
        for k in ("lm_c3", "lm_c1"):
            o = b["configs"][k]
            print(k, {n: (o[n]["ms_per_iteration"], o[n]["ms_per_iteration_by_phase"]) for n in ("gpu_device_solve", "gpu_device_solve_three_calls", "gpu_host_solve") if n in o}, o.get("error"))
PY
rm -rf /tmp/pi; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pi -o b -- python bench.py $T --no-lm > $O/rocprof_c4.log 2>&1
python scripts/r05_trace_tail.py /tmp/pi $O/with_c4_probe 30
tail -5 $O/pytest_part.txt
