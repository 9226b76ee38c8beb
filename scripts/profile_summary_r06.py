"""Summarise the round-6 evidence run (scripts/collect_evidence_r06.sh) into profiles/:   python scripts/profile_summary_r06.py gpurun_out/r06e
 * r06_bench_driver_cmd.json, r06_bench_default_2000steps.json            the bench lines themselves
 * r06_bench_driver_{cmd,noextras}_rocprofv3_kernel_stats.csv, r06_bench_2000steps_rocprofv3_kernel_stats.csv
 * r06_pmc_{fetch,write}_{20,2000}_counters.csv                           per-dispatch counters of the resident kernel (separate passes)
 * r06_hbm_stream_*                                                       the bandwidth-bound launches of roofline_hbm
 * r06_reference_api_{n}_*                                                K1' moments / K5 nl_round / K4b mask at three sizes
 * r06_bench_profiles.json    the index bench.py reads (PROFILE_INDEX): per kernel and size, rocprofv3 average launch and PMC traffic per launch
 * r06_resident_timeline.jsonl, r06_sq_counters_1M.json, r06_streaming_ab.jsonl, r06_fuzz_campaign.txt, r06_pytest_gpu.txt
PMC correction (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE reads half the bytes of a 16-B/lane coalesced stream; both counters are
in KiB.  traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024."""
import csv, glob, json, os, shutil, statistics, sys
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
RES = "normal_eq_resident_kernel<float, 0, 256, true, false, 2, 0"
CORR = "gfx950: traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (FETCH_SIZE reads half the bytes of a 16-B/lane stream; separate --pmc passes)"


def stats_rows(name, dst):
    f = os.path.join(src, name + "_kernel_stats.csv")
    if not os.path.exists(f):
        return []
    shutil.copy(f, os.path.join(prof, dst))
    return [dict(r, file="profiles/" + dst) for r in csv.DictReader(open(f))]


def pick(rows, needle):
    for r in rows:
        if needle in r["Name"]:
            return {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "file": r["file"], "name": r["Name"]}
    return None


def counters(name, counter_name, dst, needle, keep_lines=60):
    f = os.path.join(src, name + "_counters.csv")
    if not os.path.exists(f):
        return None
    rows = [r for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"] and r["Counter_Name"] == counter_name]
    if dst:
        with open(f) as fh, open(os.path.join(prof, dst), "w") as out:
            for i, line in enumerate(fh):
                if i > keep_lines:
                    break
                out.write(line)
    vals = sorted(float(r["Counter_Value"]) for r in rows)
    if not vals:
        return None
    return {"median_KiB": vals[len(vals) // 2], "dispatches": len(vals), "file": ("profiles/" + dst) if dst else None, "grid": rows[0].get("Grid_Size"), "vgpr": rows[0].get("VGPR_Count")}


sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_hash: the sources these profiles were taken on)
index = {"kernel_src_sha256": bench.kernel_source_hash(), "kernel_src_sha256_note": "sha256[:16] over " + ", ".join(bench.KERNEL_SOURCES) + " (name NUL content) of the tree the passes below were taken on; bench.py carries a file-derived traffic only while its own hash equals this one",
         "library_sha256": bench.library_hash()}
# ---- the headline's resident kernel, per steps-per-launch
entries = []
for steps, st_name, st_dst, tagn in ((20, "prof_driver_noextras", "r06_bench_driver_noextras_rocprofv3_kernel_stats.csv", "20"),
                                     (2000, "prof_2000", "r06_bench_2000steps_rocprofv3_kernel_stats.csv", "2000")):
    st = pick(stats_rows(st_name, st_dst), RES)
    fe = counters(f"pmc_fetch_{tagn}", "FETCH_SIZE", f"r06_pmc_fetch_{tagn}_counters.csv", RES)
    wr = counters(f"pmc_write_{tagn}", "WRITE_SIZE", f"r06_pmc_write_{tagn}_counters.csv", RES)
    e = {"steps_per_launch": steps}
    if st:
        e.update(kernel=st["name"], rocprofv3_avg_launch_us=st["avg_ns"] * 1e-3, rocprofv3_us_per_step=st["avg_ns"] * 1e-3 / steps, rocprofv3_calls=st["calls"], rocprofv3_file=st["file"])
    if fe and wr:
        traffic = 2 * fe["median_KiB"] * 1024 + wr["median_KiB"] * 1024
        e.update(traffic_bytes_per_launch=traffic, traffic_bytes_per_step=traffic / steps, FETCH_SIZE_KiB_median=fe["median_KiB"], WRITE_SIZE_KiB_median=wr["median_KiB"],
                 pmc_dispatches=[fe["dispatches"], wr["dispatches"]], pmc_files=[fe["file"], wr["file"]], grid_size=fe["grid"], correction=CORR)
    e["source"] = "scripts/collect_evidence_r06.sh -> scripts/profile_summary_r06.py"
    e["command"] = ("rocprofv3 ... -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-hbm" if steps == 20 else
                    "rocprofv3 ... -- python3 bench.py --steps 2000 --warmup 2000 --no-extras --no-cpu-baseline --no-hbm")
    entries.append(e)
index["normal_eq_resident_p2p_f32"] = entries
full = pick(stats_rows("prof_driver", "r06_bench_driver_cmd_rocprofv3_kernel_stats.csv"), RES)
index["driver_command_with_extras"] = None if not full else {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5", "rocprofv3_avg_launch_us": full["avg_ns"] * 1e-3, "calls": full["calls"],
    "file": full["file"], "note": "all launches of this kernel name in the run: pre-warm, host-thread tuning, timed (20 steps each), the --warmup launch (5) and the convergence leg's"}
# ---- the bandwidth-bound launches (normal_eq_kernel), told apart by grid size / bytes: 20 M, 10 M point-to-point, 1 M point-to-plane
hbm = {}
try:
    live = json.loads(open(os.path.join(src, "hbm_stream_probe.json")).read().strip().splitlines()[-1])
except Exception:
    live = {}
stats_rows("prof_hbm", "r06_hbm_stream_rocprofv3_kernel_stats.csv")
f = os.path.join(src, "prof_hbm_kernel_trace.csv")
trace = list(csv.DictReader(open(f))) if os.path.exists(f) else []


def by_case(rows, key_fn, val_fn):
    d = {}
    for r in rows:
        d.setdefault(key_fn(r), []).append(val_fn(r))
    return d


fe_f, wr_f = os.path.join(src, "pmc_fetch_hbm_counters.csv"), os.path.join(src, "pmc_write_hbm_counters.csv")
for nm, dst in ((fe_f, "r06_hbm_stream_pmc_fetch_counters.csv"), (wr_f, "r06_hbm_stream_pmc_write_counters.csv")):
    if os.path.exists(nm):
        with open(nm) as fh, open(os.path.join(prof, dst), "w") as out:
            for i, line in enumerate(fh):
                if i > 200:
                    break
                out.write(line)
# the probe launches, in order: 45 x p2p over 20 M, 45 x p2p over 10 M (kernel <float, 0, ...>), 65 x point-to-plane over 1 M (<float, 1, ...>)
def ordered(fpath, counter):
    if not os.path.exists(fpath):
        return [], []
    rows = [r for r in csv.DictReader(open(fpath)) if r["Counter_Name"] == counter]
    p2p = [float(r["Counter_Value"]) for r in rows if "normal_eq_kernel<float, 0" in r["Kernel_Name"]]
    pl = [float(r["Counter_Value"]) for r in rows if "normal_eq_kernel<float, 1" in r["Kernel_Name"]]
    return p2p, pl
fp2p, fpl = ordered(fe_f, "FETCH_SIZE")
wp2p, wpl = ordered(wr_f, "WRITE_SIZE")
def med(v):
    return statistics.median(v) if v else None
halves = lambda v: (v[: len(v) // 2], v[len(v) // 2:])
tdur = {}
if trace:
    d0 = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3 for r in trace if "normal_eq_kernel<float, 0" in r["Kernel_Name"]]
    d1 = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3 for r in trace if "normal_eq_kernel<float, 1" in r["Kernel_Name"]]
    tdur = {"p2p_20M": med(halves(d0)[0]), "p2p_10M": med(halves(d0)[1]), "config3_p2plane_1M_steady": med(d1)}
for case, alg, fv, wv in (("p2p_20M", 480e6, halves(fp2p)[0], halves(wp2p)[0]), ("p2p_10M", 240e6, halves(fp2p)[1], halves(wp2p)[1]),
                          ("config3_p2plane_1M_steady", 36e6, fpl, wpl)):
    if fv and wv:
        traffic = 2 * med(fv) * 1024 + med(wv) * 1024
        hbm[case] = {"traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": traffic / alg, "FETCH_SIZE_KiB_median": med(fv),
                     "WRITE_SIZE_KiB_median": med(wv), "pmc_dispatches": [len(fv), len(wv)], "rocprofv3_median_launch_us": tdur.get(case),
                     "event_avg_launch_us_in_profiled_run": (live.get(case) or {}).get("avg_launch_us"), "correction": CORR}
index["hbm_stream"] = hbm
index["hbm_stream_source"] = "scripts/collect_evidence_r06.sh: rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE over scripts/hbm_stream_probe.py"
# ---- the reference-API kernels, one size per profiled process
api = {}
KN = {"K1p_moments": "moments_kernel", "K5_nl_round": "nl_round", "K4b_mask_33": "mask_kernel"}
BPC = {"K1p_moments": 26, "K5_nl_round": 66, "K4b_mask_33": 26}
for n in (307200, 1000000, 10000000):
    rows = stats_rows(f"prof_api_{n}", f"r06_reference_api_{n}_rocprofv3_kernel_stats.csv")
    try:
        live = json.loads(open(os.path.join(src, f"reference_api_probe_{n}.json")).read().strip().splitlines()[-1])
    except Exception:
        live = {}
    for case, needle in KN.items():
        st = pick(rows, needle)
        fe = counters(f"pmc_fetch_api_{n}", "FETCH_SIZE", f"r06_reference_api_{n}_pmc_fetch_counters.csv", needle, keep_lines=120)
        wr = counters(f"pmc_write_api_{n}", "WRITE_SIZE", f"r06_reference_api_{n}_pmc_write_counters.csv", needle, keep_lines=120)
        e = {"n": n, "algorithmic_bytes_per_launch": BPC[case] * n}
        if st:
            e.update(kernel=st["name"], rocprofv3_avg_launch_us=st["avg_ns"] * 1e-3, rocprofv3_min_launch_us=st["min_ns"] * 1e-3, rocprofv3_calls=st["calls"], rocprofv3_file=st["file"],
                     rocprofv3_achieved_GBs=BPC[case] * n / (st["avg_ns"] * 1e-9) / 1e9, rocprofv3_frac_of_8TBs=BPC[case] * n / (st["avg_ns"] * 1e-9) / 8e12)
        if fe and wr:
            traffic = 2 * fe["median_KiB"] * 1024 + wr["median_KiB"] * 1024
            e.update(traffic_bytes_per_launch=traffic, traffic_over_algorithmic=traffic / (BPC[case] * n), FETCH_SIZE_KiB_median=fe["median_KiB"], WRITE_SIZE_KiB_median=wr["median_KiB"],
                     pmc_dispatches=[fe["dispatches"], wr["dispatches"]], pmc_files=[fe["file"], wr["file"]], correction=CORR)
        if case in live:
            e["event_avg_launch_us_in_profiled_run"] = live[case]["avg_launch_us"]
        api["%s_%d" % (case, n)] = e
index["reference_api_kernels"] = api
index["reference_api_source"] = "scripts/collect_evidence_r06.sh: rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE over scripts/reference_api_probe.py (RPE_PROBE_N = size; 35 steady launches per kernel)"
json.dump(index, open(os.path.join(prof, "r06_bench_profiles.json"), "w"), indent=1)
for name in ("bench_driver_cmd.json", "bench_default_2000steps.json", "resident_timeline.jsonl", "pytest_gpu.txt", "fuzz_campaign.txt", "gn_refine_main_20.txt", "gn_refine_main_2000.txt",
             "stream_signal.jsonl"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(prof, "r06_" + name))
for name in ("bench_rccl_world1.json", "prof_rccl_kernel_stats.csv", "bench_rccl_world1_extras.json", "kernel_roofline.jsonl", "device_loop_solver.jsonl", "device_loop_nosolver.jsonl",
             "score_probe.json", "prof_score_kernel_stats.csv", "score_filter_stats.jsonl", "session_time.json", "engine_session_on.txt", "engine_session_off.txt",
             "engine_session_phases.txt", "soak.json"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(prof, "r06_" + name.replace("prof_rccl_kernel_stats", "bench_rccl_world1_rocprofv3_kernel_stats").replace("prof_score_kernel_stats", "score_rocprofv3_kernel_stats")))
# SQ counters of the streaming kernels (medians per kernel kind)
sq = {}
for f in glob.glob(os.path.join(src, "sq", "counters_*.csv")):
    for r in csv.DictReader(open(f)):
        if "normal_eq_kernel<float, " not in r["Kernel_Name"]:
            continue
        kind = {"0": "p2p", "1": "p2plane", "2": "bearing"}.get(r["Kernel_Name"].split("normal_eq_kernel<float, ")[1][0])
        if kind:
            sq.setdefault(kind, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if sq:
    out = {"what": "SQ counters of normal_eq_kernel<float, kind, 256, false, false, CLEAN> at 1 000 000 correspondences (256 workgroups x 256 threads = 1024 waves, one per SIMD); rocprofv3 --pmc, "
                   "separate passes of three counters, medians over the dispatches; scripts/sq_pmc.sh on the round-6 tree",
           "note": "SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count in units of 4 cycles, summed over the waves"}
    for kind, d in sq.items():
        m = {c: statistics.median(v) for c, v in d.items()}
        w = m.get("SQ_WAVES", 1024.0)
        out[kind] = {"counters": m, "valu_instructions_per_wave": m.get("SQ_INSTS_VALU", 0) / w, "wave_life_cycles": 4 * m.get("SQ_WAVE_CYCLES", 0) / w,
                     "valu_active_fraction_of_wave_life": (m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_WAVE_CYCLES"]) if m.get("SQ_WAVE_CYCLES") else None}
    json.dump(out, open(os.path.join(prof, "r06_sq_counters_1M.json"), "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, (dict, list)) else "...") for k, v in index.items()}, indent=1))
