#!/bin/bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_solver_gpu.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest exit $?" >> $O/pytest.txt; tail -4 $O/pytest.txt | cut -c1-300
timeout 600 python scripts/solver_time.py > $O/solver_time.jsonl 2> $O/solver_time.err; python - <<'PY'
import json
for l in open("gpurun_out/r03k/solver_time.jsonl"):
    d=json.loads(l)
    print(d["graph"], d["poses"], "dense", d.get("dense_solve_ms"), {k.replace("sparse_","").replace("_solve_ms",""): v for k,v in d.items() if k.endswith("_solve_ms") and k.startswith("sparse")}, "vs dense", max([v for k,v in d.items() if k.endswith("vs_dense_rel")] or [0]))
PY
tail -2 $O/solver_time.err
echo "--- per-entry gather for the chains (GP_SPARSE_STAGED=0) ---"
GP_SPARSE_STAGED=0 timeout 600 python scripts/solver_time.py > $O/solver_time_v1.jsonl 2>> $O/solver_time.err; python - <<'PY'
import json
for l in open("gpurun_out/r03k/solver_time_v1.jsonl"):
    d=json.loads(l)
    print(d["graph"], d["poses"], {k.replace("sparse_","").replace("_solve_ms",""): v for k,v in d.items() if k.endswith("_solve_ms") and k.startswith("sparse")})
PY
