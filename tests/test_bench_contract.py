"""CPU: the committed bench line (profiles/r02_bench_n1.json, produced by `python bench.py` on an MI355X) carries every field of
the driver's contract, with BASELINE.json's metric, and its roofline / cpu_baseline objects are well formed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_matches_the_contract():
    line = open(os.path.join(ROOT, "profiles", "r02_bench_n1.json")).read().strip().splitlines()[-1]
    j = json.loads(line)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["metric"].split("/")[0].replace("-", " ") in base["metric"].replace("-", " ")
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    # value counts VALID correspondences (rows passing the RANSAC inlier mask), per median step time
    assert abs(j["value"] - j["config"]["valid_corr_per_step"] / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]
    assert 0.8 * j["config"]["global_corr"] < j["config"]["valid_corr_per_step"] <= j["config"]["global_corr"] == 307200
    assert j["timing"]["repeats"] >= 50 and j["timing"]["ms_per_step_p10"] <= j["ms_per_step"] <= j["timing"]["ms_per_step_p90"]
    assert j["value"] >= 1e9                                   # the north star's target on one MI355X at 307k points
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["launches_timed"] >= 16 and r["steps_per_launch"] >= 1   # bench.py --event-reps
    assert (r["traffic"] is None) == (r["traffic_source"] is None)       # a counter figure always names the profiles/ file it was read from
    assert (r["rocprofv3_avg_launch_us"] is None) == (r["rocprofv3_source"] is None)
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == j["unit"] and "sample" in c
    e = j["pose_error_vs_cpu"]
    assert e["rot_rad"] <= e["tolerance"]["rot_rad"] == 1e-5 and e["trans_rel"] <= e["tolerance"]["trans_rel"] == 1e-4
    cv = j["convergence"]                                      # iterations to |delta| < 1e-9
    for name in ("all_points", "inliers_only"):
        assert 1 <= cv[name]["iterations"] < 50 and cv[name]["last_step"] < 1e-9
        assert cv[name]["rot_rad_vs_cpu_closed_form"] <= 1e-5 and cv[name]["trans_rel_vs_cpu_closed_form"] <= 1e-4
