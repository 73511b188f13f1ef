"""bench.py's LAST stdout line is what the driver parses: one compact strict-JSON object that fits whole into the driver's 8-KB tail
(<= 4 096 bytes), with the contract's keys, `roofline` and `cpu_baseline`.  Round 5's 24-KB line left BENCH_r05.parsed = null."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "tests", "golden", "bench_full_result_r5.json")  # a full result as round 5's bench.py produced it (23.9 KB)

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline"]


def _strict(line):
    def no_const(x):
        raise ValueError("non-strict JSON constant " + x)
    return json.loads(line, parse_constant=no_const)


def test_compact_line_from_a_full_result_fits_and_round_trips():
    assert os.path.getsize(FULL) > 20000
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-line", FULL], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    lines = r.stdout.strip().splitlines()
    line = lines[-1]
    assert len(line.encode()) < 4096
    d = _strict(line)
    assert json.dumps(d, allow_nan=False, separators=(",", ":")) == line
    for k in CONTRACT:
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert set(d["configs"]) == {"c2_conn", "c1", "c5_zipf", "c3_levels"}
    for e in d["configs"].values():
        assert set(e) <= {"value", "unit", "ms_per_step", "frac", "kernel", "kernel_frac", "parity_ok", "error"}
    full = json.load(open(FULL))
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5 and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-4


def test_compact_line_never_carries_nan_and_sheds_optional_parts_when_too_long():
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(FULL))
    full["roofline"]["frac"] = float("nan")
    full["cpu_baseline"]["allcores_value"] = float("inf")
    full["configs"] = {"c%d" % i: dict(full["configs"]["c1"]) for i in range(60)}  # far more sub-runs than the line has room for
    line = bench.compact_line(bench._jsonable(full), "x.json")
    assert len(line) <= bench.LINE_LIMIT
    d = _strict(line)
    assert d["roofline"].get("frac") is None and "configs" not in d
    for k in CONTRACT:
        assert k in d
