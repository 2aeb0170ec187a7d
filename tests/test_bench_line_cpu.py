"""bench.py's LAST stdout line is what the driver records (round 5's line had grown to 20.9 KB and was not parsed -- the round counted as unmeasured).
The line is built by bench.compact_result / compact_line from the full result object; this file holds that function to the contract on CPU: every required key present,
strict JSON (no NaN / Infinity), shorter than 4 KB whatever the detail legs put into the full object, no prose in it."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _strict(line):
    def refuse(name):
        raise ValueError(f"non-finite constant {name} in the line")

    return json.loads(line, parse_constant=refuse)


def _check_contract(b):
    for key in bench.REQUIRED_KEYS:
        assert key in b, key
    for key in bench.REQUIRED_CONFIG:
        assert key in b["config"], key
    for key in bench.REQUIRED_ROOFLINE:
        assert key in b["roofline"], key
    if b["cpu_baseline"] is not None:
        for key in bench.REQUIRED_CPU:
            assert key in b["cpu_baseline"], key
    assert b["vs_baseline"] is None and b["higher_is_better"] is True and b["scaling"] == "weak" and b["data"] == "synthetic"
    assert b["detail_file"] == "bench_detail.json"

    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
        elif isinstance(x, str):
            yield x

    assert max(len(s) for s in strings(b)) <= 120  # names, not notes


def _fat_full_result():
    """the largest full object on record (round 5's 20.9 KB line: every leg, every note) dressed as this round's full result, plus non-finite values"""
    full = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r05_bench_run1.json")) if l.startswith("{")][-1])
    full["roofline"].update(frac_rocprof=0.47, rocprof_avg_ms=0.0148, rocprof_calls=10031, rocprof_detail=dict(source="x" * 400))
    full["config"].update(exchange=None, device_warmup_ms=300.0, device_warmup_steps=13248)
    full["parity_vs_oracle"]["H_target"] = float("nan")
    full["configs"]["C3"]["parity_vs_reference_max"] = float("inf")
    full["legs_skipped"] = ["c4"]
    full["run_seconds"] = 44.0
    full["exchange_ms"] = dict(all_reduce=0.061, all_gather=0.055, peer=None)
    full["exchange_verified"], full["rccl_world"], full["backend"] = True, 8, "nccl"
    return full


def test_compact_line_of_the_fattest_result_is_short_strict_and_complete():
    full = _fat_full_result()
    assert len(json.dumps(full)) > 15000  # (the object that broke round 5's record)
    line = bench.compact_line(full)
    assert "\n" not in line and len(line.encode()) < bench.MAX_LINE_BYTES == 4096
    assert len(line.encode()) < 3000  # head-room: the legs may grow a few numbers, not notes
    b = _strict(line)
    _check_contract(b)
    assert b["value"] == full["value"] and b["ms_per_step"] == full["ms_per_step"] and b["ms_per_step_cold"] == full["ms_per_step_cold"]
    assert b["roofline"]["frac"] == full["roofline"]["frac"] and b["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert b["cpu_baseline"]["kind"] == "reference" and b["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert b["parity_max"] is None and b["parity_ok"] is False  # (a NaN entry is a failed parity, never a hidden one)
    full["parity_vs_oracle"]["H_target"] = 3e-9
    b2 = _strict(bench.compact_line(full))
    assert b2["parity_ok"] is True and 0 < b2["parity_max"] < 1e-5
    assert b["legs"]["C3_ms"] == full["configs"]["C3"]["ms"] and b["legs"]["C3_parity"] is None  # (non-finite -> null)
    assert b["legs"]["map_build_ms"] == full["configs"]["map_build"]["ms"] and b["legs"]["big_source_frac"] == full["big_source"]["roofline"]["frac"]
    assert b["exchange_verified"] is True and b["rccl_world"] == 8 and b["exchange_ms"]["peer"] is None


def test_compact_line_of_a_bare_result_still_carries_every_key():
    """a run with every optional leg skipped and no CPU baseline (N > 1, rank 0): the keys are there, the values null"""
    full = dict(metric=bench.METRIC, value=1.0, unit="point-correspondences/s", n_gpus=2, steps=20, warmup=5, ms_per_step=0.05, ms_per_step_cold=None, higher_is_better=True,
                scaling="weak", vs_baseline=None, dtype="f64", data="synthetic", config=dict(workload="w"), roofline=dict(bound="hbm"), cpu_baseline=None, parity_vs_oracle=None)
    b = _strict(bench.compact_line(full))
    for key in bench.REQUIRED_KEYS:
        assert key in b, key
    assert b["cpu_baseline"] is None and b["parity_max"] is None and b["legs"] == {} and set(bench.REQUIRED_ROOFLINE) <= set(b["roofline"])


def test_a_line_that_would_not_fit_is_refused():
    full = _fat_full_result()
    full["legs_skipped"] = ["leg%04d" % i for i in range(600)]
    with pytest.raises(RuntimeError, match="bytes"):
        bench.compact_line(full)


def test_recorded_lines_of_this_round_meet_the_contract():
    """every bench line committed under profiles/ this round (the builder's runs of the driver's command) is one the driver can parse"""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_bench_run*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r06_bench_rehearsal*.json")))
    assert paths, "the round's bench lines are committed under profiles/"
    for path in paths:
        lines = [l for l in open(path) if l.startswith("{")]
        if not lines:
            continue
        line = lines[-1].rstrip("\n")
        assert len(line.encode()) < 4096, path
        b = _strict(line)
        _check_contract(b)
        if b["n_gpus"] == 1:
            r = b["roofline"]
            assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["bound"] == "hbm" and r["kernel"] == "vgicp_stream_kernel"
            assert b["cpu_baseline"]["kind"] in ("reference", "port") and b["cpu_baseline"]["cores"] >= 1
            assert b["parity_max"] < 1e-5 and b["parity_inliers_equal"] is True and b["parity_ok"] is True
            assert r["frac_rocprof"] is not None and r["traffic"] is not None  # (measured in the run: rocprofv3 child passes)
        else:  # the N > 1 line: the contract's collective, every form timed and verified, rank 0's row held to the oracle
            assert b["config"]["exchange"] == "all_reduce" and b["exchange_verified"] is True and set(b["exchange_ms"]) == {"all_reduce", "all_gather", "peer"}
            assert b["rccl_world"] == b["n_gpus"] and b["parity_ok"] is True and b["cpu_baseline"] is None


def test_last_resort_line_is_printed_when_a_phase_expires_and_only_then(tmp_path):
    """an optional N > 1 phase that hangs on hardware nobody could test on must not cost the job its headline: rank 0 registers the compact line once the headline is
    measured; a phase guard that expires (or a SIGTERM from the launcher) prints it -- marked `aborted` -- as the last stdout line; a normal end prints nothing extra"""
    import subprocess
    import textwrap

    full = dict(metric=bench.METRIC, value=1.0, unit="point-correspondences/s", n_gpus=2, steps=20, warmup=5, ms_per_step=0.05, ms_per_step_cold=None, higher_is_better=True, scaling="weak",
                vs_baseline=None, dtype="f64", data="synthetic", config=dict(workload="w", exchange="all_reduce"), roofline=dict(bound="hbm"), cpu_baseline=None, parity_vs_oracle=None)
    line = bench.compact_line(full)
    prog = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench_detail
        bench_detail.register_last_resort({line!r})
        mode = sys.argv[1]
        if mode == "expire":
            with bench_detail.PhaseGuard(0.3, "peer exchange"):
                time.sleep(5)
        elif mode == "term":
            import os, signal
            os.kill(os.getpid(), signal.SIGTERM)
            time.sleep(5)
        else:
            bench_detail.register_last_resort(None)
            print("the real line")
    """)
    f = tmp_path / "p.py"
    f.write_text(prog)
    for mode, rc in (("expire", 124), ("term", 143)):
        p = subprocess.run([sys.executable, str(f), mode], capture_output=True, text=True, timeout=60)
        assert p.returncode == rc, (mode, p.returncode, p.stderr[-300:])
        b = _strict(p.stdout.rstrip("\n").splitlines()[-1])
        assert b["value"] == 1.0 and b["n_gpus"] == 2 and "aborted" in b and len(p.stdout.strip().splitlines()) == 1
    p = subprocess.run([sys.executable, str(f), "normal"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout.strip() == "the real line"
