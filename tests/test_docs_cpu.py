"""The numbers quoted in DESIGN.md / README.md / BASELINE.md / docs/HISTORY.md are generated from the committed profiles/ files (scripts/r06_numbers.py --write pastes round 6's, scripts/r05_numbers.py --write round 5's between
<!-- r05:NAME:begin/end --> markers; round 4's blocks that DESIGN.md keeps for comparison come from scripts/r04_numbers.py): this test regenerates the blocks and holds the
documents to them, so that a number cannot be typed by hand or go stale behind a new evidence run."""
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _check(tag, gen):
    blocks = {name: fn() for name, fn in gen.SECTIONS.items()}
    seen = set()
    for doc in ("DESIGN.md", "README.md", "BASELINE.md", "docs/HISTORY.md"):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        for name, want in blocks.items():
            for m in re.finditer(r"<!-- %s:%s:begin -->\n(.*?)\n<!-- %s:%s:end -->" % (tag, name, tag, name), text, re.S):
                assert m.group(1) == want, f"{doc}: block {tag}:{name} differs from what scripts/{tag}_numbers.py generates from profiles/{tag}_* (run it with --write)"
                seen.add((doc, name))
    return seen


def test_documents_quote_the_committed_profiles():
    seen = _check("r05", _generator("r05_numbers"))
    for need in [("docs/HISTORY.md", "headline"), ("docs/HISTORY.md", "results"), ("docs/HISTORY.md", "lm"), ("docs/HISTORY.md", "c5"), ("BASELINE.md", "results"), ("BASELINE.md", "headline"), ("BASELINE.md", "lm")]:
        assert need in seen, need
    seen4 = _check("r04", _generator("r04_numbers"))  # (the round-4 tables: docs/HISTORY.md since round 6)
    assert ("docs/HISTORY.md", "headline") in seen4 and ("docs/HISTORY.md", "results") in seen4
    seen6 = _check("r06", _generator("r06_numbers"))  # this round
    for need in [("DESIGN.md", "headline"), ("DESIGN.md", "results"), ("DESIGN.md", "lm"), ("DESIGN.md", "solver"), ("README.md", "results"), ("BASELINE.md", "results"), ("BASELINE.md", "headline")]:
        assert need in seen6, need


def test_the_bench_line_in_profiles_meets_the_contract():
    for path in ("r05_bench_run1.json", "r05_bench_run2.json", "r05_bench_run3.json"):
        line = [l for l in open(os.path.join(ROOT, "profiles", path)) if l.startswith("{")][-1]
        b = json.loads(line)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_cold", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
                    "cpu_baseline"):
            assert key in b, key
        assert b["steps"] == 20 and b["warmup"] == 5 and b["n_gpus"] == 1  # the driver's command
        r = b["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_streaming", "frac_fused_kernel", "actual_bytes", "frac_actual", "cold"):
            assert key in r, key
        # frac is the WHOLE fused kernel (== frac_fused_kernel), not its streaming slice; traffic was measured in the run
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["frac"] == r["frac_fused_kernel"] and r["frac"] < r["frac_streaming"]
        assert r["traffic"] and "measured in THIS run" in r["traffic_source"] and 0.5 * r["actual_bytes"] < r["traffic"] < 1.5 * r["actual_bytes"]
        assert b["vs_baseline"] is None and "workload" in b["config"]
        c = b["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and "sample" in c
        cfg = b["configs"]
        assert "map_build" in cfg and "big_source" in b and b["c4"]["exchange"] in ("none", "all_gather", "all_reduce")
        for key in ("lm_c3", "lm_c1"):  # the optimizer iteration, with the gate
            o = cfg[key]
            assert "error" not in o, o
            for leg in ("gpu_device_solve", "gpu_host_solve", "cpu_baseline"):
                assert o[leg]["gate_met"] and set(o[leg]["ms_per_iteration_by_phase"]) == {"linearize", "solve", "error", "glue"}
            assert o["gpu_device_solve"]["iterations"] == o["cpu_baseline"]["iterations"]
        assert cfg["C5"]["covariances"]["ms_kitti_scan"] is not None
