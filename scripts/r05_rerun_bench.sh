set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_run$i.log 2>&1; grep "^{" $O/bench_run$i.log > $O/bench_run$i.json; done
cp $O/bench_run1.json $O/bench_n1.json; grep -h "^real" $O/bench_run*.log
python - <<'PY'
import json
for i in (1,2,3):
    b=json.loads(open(f"gpurun_out/r05/bench_run{i}.json").read())
    c=b["configs"]["C5"]["covariances"]
    print(i, b["ms_per_step"], b["roofline"]["frac"], c["ms"], c["ms_target_cloud"], c["ms_kitti_scan"], b["configs"]["C3"]["cpu_baseline"]["ms"], b["configs"]["C3"]["cpu_baseline"]["cores"], b["cpu_baseline"]["cores"], b["cpu_baseline"]["ms_per_linearize"], b["configs"]["lm_c3"]["cpu_baseline"]["ms_per_iteration"], b["configs"]["lm_c3"]["gpu_device_solve"]["ms_per_iteration"])
PY
