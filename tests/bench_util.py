"""Test helper: run bench.py as the driver does and return (process, the ONE stdout line parsed, the full record of bench_extras.json).
The line is checked here against the driver's limits: one JSON line, last on stdout, under 4 KB, the contract's keys."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "avg_launch_us", "steps_per_launch")


def check_line(text):
    assert "\n" not in text and len(text) < 4096, len(text)
    j = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in j, k
    for k in ROOFLINE_KEYS:
        assert k in j["roofline"], k
    assert "workload" in j["config"] and "model" not in j["config"]
    assert all(not isinstance(v, (dict, list)) for o in (j["config"], j["roofline"], j["cpu_baseline"] or {}) for v in o.values())   # flat objects
    return j


def run_bench(argv, env, timeout=900, expect_rc=0):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "extras.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=dict(env, RPE_BENCH_EXTRAS=path), capture_output=True, text=True,
                           timeout=timeout)
        assert r.returncode == expect_rc, (r.returncode, r.stdout[-800:], r.stderr[-2500:])
        line = check_line(r.stdout.strip().splitlines()[-1])
        full = json.load(open(path))
    assert full["value"] == line["value"] or abs(full["value"] - line["value"]) < 1e-6 * abs(full["value"])
    return r, line, full
