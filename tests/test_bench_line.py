"""The driver keeps about 2,000 characters of bench.py's stdout and parses the LAST line: round 2's 21 KB line was
recorded as `parsed: null`.  The formatter must keep every contract key inside that budget whatever the legs carry."""
import copy
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned():
    with open(os.path.join(ROOT, "profiles", "r02_bench_driver_form.json")) as f:     # a real full document (21 KB)
        return json.load(f)


def test_compact_line_of_a_real_document_fits_and_round_trips():
    full = _canned()
    assert len(json.dumps(full)) > 15000
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in line and len(line) < 2048 and len(line) <= bench.COMPACT_LIMIT
    doc = json.loads(line)
    for key in CONTRACT:
        assert key in doc, key
    assert doc["config"]["workload"].startswith("C3: 10000000")
    assert doc["config"]["name"] == "C3" and doc["config"]["queries_per_step"] == 256
    assert abs(doc["value"] - full["value"]) < 1e-3 * full["value"]
    assert abs(doc["ms_per_step"] - full["ms_per_step"]) < 1e-3 * full["ms_per_step"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step",
                "algo_bytes_per_launch", "frac_survey_8d"):
        assert key in doc["roofline"], key
    assert abs(doc["roofline"]["frac"] - doc["roofline"]["achieved"] / doc["roofline"]["peak"]) < 1e-3
    for key in ("value", "unit", "cores", "kind", "sample", "threads_used", "os_cpu_count", "dotnet"):
        assert key in doc["cpu_baseline"], key
    assert doc["parity"]["rank_identical"] is True and doc["parity"]["max_abs_score_delta"] == 0.0
    assert doc["parity"]["rows_checked"] == 320


def test_compact_line_stays_short_under_hostile_documents():
    full = _canned()
    big = copy.deepcopy(full)
    for i in range(40):                                    # far more legs than fit: the legs go, the contract keys stay
        big["legs"]["extra_leg_number_%d_with_a_long_name" % i] = copy.deepcopy(full["legs"]["C2_1M_rows_1_query"])
    big["cpu_baseline"]["sample"] = "x" * 5000
    big["error"] = "RuntimeError: " + "y" * 5000
    big["n_gpus"] = 8
    big["config"]["parallelism"] = "z" * 3000
    line = bench.compact_line(big, "bench_full.json")
    assert len(line) < 2048
    doc = json.loads(line)
    for key in CONTRACT:
        assert key in doc, key
    assert doc["error"].startswith("RuntimeError")
    none = bench.compact_line({"metric": "m", "value": float("nan"), "roofline": None, "cpu_baseline": None, "config": {}})
    assert json.loads(none)["value"] is None               # NaN is not JSON


def test_emit_prints_exactly_one_last_line(tmp_path, capsys, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    bench.emit(_canned())
    out = capsys.readouterr().out
    assert out.endswith("\n") and out.count("\n") == 1 and len(out) < 2048
    with open(tmp_path / "gpurun_out" / "bench_full.json") as f:
        assert json.load(f)["config"]["name"] == "C3"
    assert json.loads(out)["full"] == os.path.join("gpurun_out", "bench_full.json")
