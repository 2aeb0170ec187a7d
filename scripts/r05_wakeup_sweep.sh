# Round 5: does the untimed device wake-up in front of the W warm-up steps still pay, and how long should it be?  The driver's K = 20 / W = 5 headline (no other legs) with
# --device-warmup-ms 0 / 30 / 100 / 300, four alternating rounds on one box: ms_per_step, the kernel's own whole-kernel time, and the cold leg of the same process beside them.
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05w; mkdir -p $O
cd $GRAFT_REPO_ROOT
F="--gpus 1 --steps 20 --warmup 5 --no-configs --no-c4 --no-traffic --no-big-source --no-cpu-baseline"
for rep in 1 2 3 4; do for ms in 300 0 30 100; do
  timeout 120 python bench.py $F --device-warmup-ms $ms 2>/dev/null | grep '^{' | python -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']
print(json.dumps(dict(wakeup_ms=$ms, rep=$rep, ms_per_step=b['ms_per_step'], kernel_us=round(r['kernel_ms']*1e3,2), cold_ms_per_step=b['ms_per_step_cold'], cold_kernel_us=(r.get('cold') or {}).get('fused_kernel_us'))))" >> $O/sweep.jsonl
done; done
cat $O/sweep.jsonl
