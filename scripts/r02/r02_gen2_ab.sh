#!/bin/bash
# A/B of the second-generation tile kernel (variants 9 / 10, gp_vgicp_tile2.hpp) against the round-2 default (8) on the GPU box:
# parity of every variant on the fixture, C2 (1 M and 8 M source points) tile-kernel time + per-workgroup timeline, C3 / C4-shard time.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vgicp_gpu.py -q -x -k "every_kernel_variant or second_generation or determinism" > gpurun_out/gen2_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/gen2_pytest.log
tail -5 gpurun_out/gen2_pytest.log
timeout 600 python scripts/r02_sweep.py ${GEN2_VARIANTS:-8,9,10,11} 0 --big > gpurun_out/gen2_sweep.jsonl 2> gpurun_out/gen2_sweep.err; echo "sweep exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/gen2_sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    if "variant" in d and "tile_ms" in d: print(d["case"], "v", d["variant"], "tile", d["tile_ms"], "pass", d["pass_ms"], "sync", d["sync_call_ms"], "frac", d["frac"], "err", d["max_rel_err"])
    if "trace" in d: print(d["trace"], d["phase_median_us"], d["device_axis"]["end_p50_us"], d["device_axis"]["end_max_us"])
PY
GP_VARIANTS=${GEN2_VARIANTS:-8,9,10,11} timeout 900 python scripts/bench_configs.py C3,C4 > gpurun_out/gen2_configs.jsonl 2> gpurun_out/gen2_configs.err; echo "configs exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/gen2_configs.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["config"][:60], "...", d["config"][-12:], "ms", d["ms_per_linearize"], "tile", d["tile_kernel_ms"], "frac", d["roofline_frac"], "err", d["parity_max_rel_err"])
PY
