"""The bench line the driver parses stays compact (round 4's 20 kB line came back `parsed: null`).
Builds the headline dict from RECORDED full results (profiles/*.json hold what bench_detail.json holds) -- no GPU."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
RECORDED = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r0[2-9]_bench_*.json")) if not p.endswith("_line.json"))   # full records
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_bench_*_line.json")))                                            # compact lines as printed


def _reject_constant(c):
    raise AssertionError(f"non-strict JSON constant {c}")


@pytest.mark.parametrize("path", RECORDED, ids=[os.path.basename(p) for p in RECORDED])
def test_compact_line_from_recorded_detail(path):
    res = json.load(open(path))
    if "metric" not in res:
        pytest.skip("not a bench record")
    line = json.dumps(bench.compact_line(res), allow_nan=False)
    assert len(line) < bench.LINE_BUDGET <= 6000, len(line)
    back = json.loads(line, parse_constant=_reject_constant)
    for k in REQUIRED:
        assert k in back, k
    assert back["value"] == res["value"] and back["ms_per_step"] == res["ms_per_step"]       # the headline keeps every digit
    assert "roofline_all" not in back and "configs_measured" not in back
    if "roofline" in res:
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in back["roofline"], k
        assert back["roofline"]["frac"] == pytest.approx(res["roofline"]["frac"], rel=1e-3)
        src = back["roofline"].get("traffic_source")
        assert src is None or (src.endswith(".json") and " " not in src)
    if "cpu_baseline" in res:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k
    for v in back["config"].values():
        assert not isinstance(v, (dict, list)) or len(json.dumps(v)) < 200


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_recorded_compact_lines_are_what_the_driver_can_parse(path):
    raw = open(path).read().strip()
    assert "\n" not in raw and len(raw) < bench.LINE_BUDGET
    line = json.loads(raw, parse_constant=_reject_constant)
    for k in REQUIRED:
        assert k in line, k
    assert "roofline_all" not in line and "configs_measured" not in line and line["detail"] == "bench_detail.json"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])


def test_recorded_files_cover_both_workloads():
    names = [os.path.basename(p) for p in RECORDED]
    assert any("default" in n for n in names) and any("mmarco" in n or "launcher" in n for n in names)


def test_non_finite_floats_become_null():
    res = {"metric": "m", "value": 1.0, "ms_per_step": 1.0, "config": {"workload": "w", "x": float("nan")},
           "stages_ms": {"a": float("inf")}, "roofline": {"kernel": "k", "frac": 0.5, "traffic": None, "traffic_source": None}}
    line = json.dumps(bench.compact_line(res), allow_nan=False)
    back = json.loads(line)
    assert back["config"]["x"] is None and back["stages_ms"]["a"] is None


def test_emit_prints_one_stdout_line_and_writes_the_detail(tmp_path, capsys, monkeypatch):
    res = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(res)
    out, err = capsys.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"') and len(lines[0]) < 6000
    assert json.load(open(tmp_path / "bench_detail.json"))["configs_measured"] == res["configs_measured"]
    assert "[bench detail]" in err
