"""CPU: the emitter of the CURRENT bench.py (contract_line) run on a canned full record -- the round-4 record (21.9 KB as one line, which
the driver could not parse) -- must give one JSON line under 4 KB with exactly the contract's keys, flat objects and BASELINE.json's metric;
whatever else a run measured belongs in bench_extras.json (write_extras), never in the line."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports nothing heavy at module level: no torch, no HIP library)
from bench_util import check_line  # noqa: E402


def canned():
    return json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_driver_cmd.json")).read().strip().splitlines()[-1])


def test_line_of_the_current_emitter_is_small_flat_and_complete():
    full = canned()
    assert len(json.dumps(full)) > 20000                     # the record that broke the driver's parser
    text = bench.contract_line(full)
    j = check_line(text)                                     # one line, < 4096 bytes, contract keys, flat config / roofline / cpu_baseline
    assert len(text) < 2500
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"].split("/")[0].replace("-", " ") in base["metric"].replace("-", " ")
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "configs[1]" in j["config"]["workload"]
    assert abs(j["value"] - j["config"]["valid_corr_per_step"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"] and j["value"] >= 1e9
    assert j["config"]["ms_per_step_p10"] <= j["ms_per_step"] <= j["config"]["ms_per_step_p90"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-4 * r["achieved"]
    assert r["bytes_per_launch"] == 26 * 307200 * r["steps_per_launch"] and r["steps_per_launch"] == 20
    assert abs(r["traffic_over_algorithmic"] - r["traffic"] / r["bytes_per_launch"]) < 1e-5
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == j["unit"] and c["sample"] and c["all_cores_value"] > 0
    e = j["pose_error_vs_cpu"]
    assert e["rot_rad"] <= e["tol_rot_rad"] == 1e-5 and e["trans_rel"] <= e["tol_trans_rel"] == 1e-4
    extra = set(j) - {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                      "config", "roofline", "cpu_baseline", "pose_error_vs_cpu", "extras_file"}
    assert extra <= {"config3_frac_steady", "config3_frac_cold", "k4_exact_33_valu_frac"}            # at most three scalar extras
    assert all(isinstance(j[k], float) for k in extra)


def test_growth_of_the_full_record_never_reaches_the_line():
    full = canned()
    full["roofline_hbm"]["reference_api_kernels"].update({f"more_{i}": {"note": "x" * 500} for i in range(100)})
    full["roofline"]["note"] = "y" * 10000
    full["anything_new"] = {"z": list(range(5000))}
    assert len(bench.contract_line(full)) < 2500


def test_missing_legs_and_odd_values_still_give_a_valid_line():
    full = canned()
    for k in ("cpu_baseline", "roofline_hbm", "pose_error_vs_cpu", "ransac_scoring", "timing"):
        full.pop(k, None)
    full["cpu_baseline"] = None
    full["roofline"]["traffic"] = None
    full["roofline"]["traffic_over_algorithmic"] = float("nan")
    j = json.loads(bench.contract_line(full))
    assert j["cpu_baseline"] is None and j["roofline"]["traffic"] is None and j["roofline"]["traffic_over_algorithmic"] is None
    full["valid"], full["invalid_reason"] = False, "3 resident grid(s) lost during the run " * 20
    j = json.loads(bench.contract_line(full))
    assert j["valid"] is False and len(j["invalid_reason"]) <= 160


def test_a_line_over_the_limit_degrades_and_only_the_strict_form_refuses(monkeypatch):
    """After a whole measurement the driver must get ONE valid line whatever was added to the record: over the limit the emitter drops the
    optional scalars, shortens the strings and, last, keeps the contract's own members only; strict=True (this test) refuses instead."""
    full = canned()
    full["cpu_baseline"]["cpu_model"] = "a very long CPU model string " * 40
    full["cpu_baseline"]["sample_short"] = "s" * 4000
    monkeypatch.setattr(bench, "LINE_LIMIT", 1400)
    with pytest.raises(RuntimeError):
        bench.contract_line(full, strict=True)
    text = bench.contract_line(full)
    assert len(text) < 1400 and "\n" not in text
    j = json.loads(text)
    assert j["truncated"] is True
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in j
    assert j["config"]["workload"] and j["roofline"]["bound"] == "hbm" and j["roofline"]["frac"] > 0 and j["cpu_baseline"]["cores"] == 1
    monkeypatch.setattr(bench, "LINE_LIMIT", 700)            # even the last resort fits
    j = json.loads(bench.contract_line(full))
    assert len(bench.contract_line(full)) < 700 and j["value"] >= 1e9 and set(j["roofline"]) == {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_file_derived_traffic_names_its_source_and_goes_null_when_the_kernel_changed():
    """roofline.traffic comes from a committed PMC profile: the line says which file and which kernel-source hash, and bench.py keeps it
    only while that hash is the hash of the sources this build was compiled from."""
    idx = bench.profile_entries()
    assert idx.get("kernel_src_sha256"), "the profile index must record the kernel sources its PMC passes were taken on"
    assert bench.kernel_source_hash() and len(bench.kernel_source_hash()) == 16
    full = canned()
    full["roofline"]["traffic_source"] = "profiles/x.json (PMC passes of this command; kernel sources abc = this build)"
    assert json.loads(bench.contract_line(full))["roofline"]["traffic_source"].startswith("profiles/x.json")
    full["roofline"].pop("traffic_source")
    assert json.loads(bench.contract_line(full))["roofline"]["traffic_source"] == "none"


def test_extras_file_holds_the_full_record(tmp_path, monkeypatch):
    monkeypatch.setenv("RPE_BENCH_EXTRAS", str(tmp_path / "x.json"))
    full = canned()
    bench.write_extras(full)
    assert json.load(open(tmp_path / "x.json")) == full


def test_sharded_record_check_judges_every_entry_on_its_own_scale():
    import numpy as np
    rng = np.random.default_rng(0)
    J = rng.standard_normal((500, 6)) * np.array([1, 1, 1, 30, 30, 30])     # rotation columns ~ |p|: H spans three orders of magnitude
    r = 0.05 * rng.standard_normal(500)
    H, g = J.T @ J, J.T @ r
    rec = np.zeros(32); k = 0
    for i in range(6):
        for j in range(i, 6):
            rec[k] = H[i, j]; k += 1
    rec[21:27], rec[27], rec[28] = g, r @ r, 500
    assert bench.record_close(rec * (1 + 1e-7), rec, 2e-6)
    bad = rec.copy(); bad[21] *= 1.5                  # a wrong gradient entry: tiny against max |H|, caught on its own scale
    assert np.max(np.abs(bad[:29] - rec[:29])) <= 2e-6 * np.max(np.abs(rec[:29])) and not bench.record_close(bad, rec, 2e-6)
    half = rec * 0.5                                   # a missing shard
    assert not bench.record_close(half, rec, 2e-6)
