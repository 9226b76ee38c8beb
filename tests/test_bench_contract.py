"""CPU: the committed bench line (profiles/r01_bench_n1.json, produced by `python bench.py` on an MI355X) carries every field of
the driver's contract, with BASELINE.json's metric, and its roofline / cpu_baseline objects are well formed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_matches_the_contract():
    line = open(os.path.join(ROOT, "profiles", "r01_bench_n1.json")).read().strip().splitlines()[-1]
    j = json.loads(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["metric"].split("/")[0].replace("-", " ") in base["metric"].replace("-", " ")
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - j["config"]["global_corr"] * j["steps"] / (j["ms_per_step"] * 1e-3 * j["steps"])) < 1e-6 * j["value"]
    assert j["value"] >= 1e9                                   # the north star's target on one MI355X at 307k points
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or 0.9 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 1.5 * r["algorithmic_bytes_per_launch"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == j["unit"] and "sample" in c
    e = j["pose_error_vs_cpu"]
    assert e["rot_rad"] <= e["tolerance"]["rot_rad"] == 1e-5 and e["trans_rel"] <= e["tolerance"]["trans_rel"] == 1e-4
