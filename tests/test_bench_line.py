"""The bench contract's ONE stdout line: compact, parseable, last (round 4's line was 20 KB and the driver's tail buffer cut it:
BENCH_r04.parsed == null).  CPU tests over full records committed under profiles/ by earlier rounds."""

import glob
import io
import json
import os
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[34]_*_bench_driver_cmd.json")) +
                 glob.glob(os.path.join(ROOT, "profiles", "r04_*_bench_c2.json")))


def _bench():
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


@pytest.mark.parametrize("path", RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_compact_line_is_small_and_complete(path):
    bench = _bench()
    full = json.load(open(path))
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT < 8192, len(text)
    assert "\n" not in text
    back = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in back, key
    assert "workload" in back["config"] and "model" not in back["config"]
    roof = back["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes_per_launch"):
        assert key in roof, key
    # the top-level roofline fields are ONE reading: they recompute from each other
    assert roof["achieved"] == pytest.approx(roof["algorithmic_bytes_per_launch"] / roof["avg_launch_us"] * 1e-3, rel=2e-3)
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=2e-4)
    assert roof["achieved"] <= roof["peak"]
    cpu = back["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]


def test_emit_prints_one_line_last_and_writes_the_full_record(tmp_path, monkeypatch):
    bench = _bench()
    full = json.load(open(RECORDS[-1]))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full)
    out = buf.getvalue()
    assert out.endswith("\n") and out.count("\n") == 1  # nothing before, nothing after
    assert json.loads(out)["value"] == full["value"]
    assert json.load(open(tmp_path / "bench_full.json")) == full


def test_compact_line_sheds_optional_objects_before_it_outgrows_the_limit():
    bench = _bench()
    full = json.load(open(RECORDS[-1]))
    full["strong_scaling"] = {f"k{i}": "x" * 70 for i in range(200)}
    full["n_gpus"] = 8
    line = bench.compact_line(full)
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line


def test_ik_reference_protocol_rows_reach_the_line():
    """the six rows of the reference's IK table travel in the line as [IK, collision-free IK] per robot, next to the published values"""
    bench = _bench()
    full = json.load(open(RECORDS[-1]))
    rows = []
    for i, robot in enumerate(("franka", "dual_ur10e", "unitree_g1")):
        for cfree in (False, True):
            rows.append({"robot": robot, "collision_free": cfree, "ms": 1.0 + i + 0.5 * cfree, "success_percent": 100.0 - cfree,
                         "published_ms_nvidia": bench.IK_PROTOCOL_PUBLISHED_MS[robot][int(cfree)]})
    rows[3] = {"robot": "dual_ur10e", "collision_free": True, "error": "RuntimeError: x"}  # a failed row leaves a gap, not a crash
    full["ik_reference_protocol"] = {"rows": rows}
    line = bench.compact_line(full)
    proto = line["ik"]["reference_protocol"]
    assert proto["ms"] == {"franka": [1.0, 1.5], "dual_ur10e": [2.0, None], "unitree_g1": [3.0, 3.5]}
    assert proto["published_ms"]["unitree_g1"] == [31.39, 526.9] and proto["success_percent"]["franka"] == [100.0, 99.0]
    assert len(json.dumps(line)) < bench.LINE_LIMIT
