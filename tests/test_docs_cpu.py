"""The round-4 numbers quoted in DESIGN.md / README.md / BASELINE.md are generated from the committed profiles/r04_* files (scripts/r04_numbers.py --write pastes them
between <!-- r04:NAME:begin/end --> markers): this test regenerates the blocks and holds the documents to them, so that a number cannot be typed by hand or go stale
behind a new evidence run."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("r04_numbers", os.path.join(ROOT, "scripts", "r04_numbers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_documents_quote_the_committed_profiles():
    gen = _generator()
    blocks = {name: fn() for name, fn in gen.SECTIONS.items()}
    seen = set()
    for doc in ("DESIGN.md", "README.md", "BASELINE.md"):
        text = open(os.path.join(ROOT, doc), encoding="utf-8").read()
        for name, want in blocks.items():
            for m in re.finditer(r"<!-- r04:%s:begin -->\n(.*?)\n<!-- r04:%s:end -->" % (name, name), text, re.S):
                assert m.group(1) == want, f"{doc}: block r04:{name} differs from what scripts/r04_numbers.py generates from profiles/r04_* (run it with --write)"
                seen.add((doc, name))
    assert ("DESIGN.md", "headline") in seen and ("DESIGN.md", "results") in seen and ("README.md", "results") in seen and ("BASELINE.md", "results") in seen


def test_the_bench_line_in_profiles_meets_the_contract():
    import json

    line = [l for l in open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")) if l.startswith("{")][-1]
    b = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    r = b["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_fused_kernel", "actual_bytes", "frac_actual"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and b["vs_baseline"] is None and "workload" in b["config"]
    c = b["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and "sample" in c
    assert "map_build" in b["configs"] and "big_source" in b and b["c4"]["exchange"] in ("none", "all_gather", "all_reduce")
