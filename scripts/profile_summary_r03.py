"""Summarise the round-3 profiles of bench.py (scripts/collect_evidence_r03.sh) into profiles/:
    python scripts/profile_summary_r03.py gpurun_out/r03b
 * r03_bench_driver_cmd_rocprofv3_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --gpus 1 --steps 20 --warmup 5`
 * r03_bench_driver_noextras_rocprofv3_kernel_stats.csv   the same with --no-extras --no-cpu-baseline --no-hbm (only the timed kind of launch)
 * r03_bench_2000steps_rocprofv3_kernel_stats.csv
 * r03_pmc_{fetch,write}_{20,2000}_counters.csv          per-dispatch FETCH_SIZE / WRITE_SIZE of the resident kernel (separate passes)
 * r03_bench_profiles.json    the index bench.py reads: per steps-per-launch, the rocprofv3 average launch and the PMC traffic per launch
PMC correction (MI355X_MICROARCH.md, HBM): on gfx950 FETCH_SIZE reads half the bytes of a 16-B/lane coalesced stream; both counters are
in KiB.  traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024."""
import csv, json, os, shutil, sys
src = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
KERNEL = "normal_eq_resident_kernel<float, 0, 512, true, false, true, false>"


def stats(name, dst):
    f = os.path.join(src, name + "_kernel_stats.csv")
    if not os.path.exists(f):
        return None
    shutil.copy(f, os.path.join(prof, dst))
    for row in csv.DictReader(open(f)):
        if KERNEL in row["Name"]:
            return {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]), "file": "profiles/" + dst}
    return None


def counter(name, counter_name, dst, steps):
    f = os.path.join(src, name + "_counters.csv")
    if not os.path.exists(f):
        return None
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter_name]
    with open(f) as fh, open(os.path.join(prof, dst), "w") as out:
        for i, line in enumerate(fh):   # head of the per-dispatch file as committed evidence
            if i > 80:
                break
            out.write(line)
    vals = [float(r["Counter_Value"]) for r in rows]
    if not vals:
        return None
    # the --warmup launch serves a different number of steps (5 against 20) and the first launch reads the arrays cold: the MEDIAN
    # dispatch is a timed-length launch in steady state
    vals.sort()
    return {"median_KiB": vals[len(vals) // 2], "dispatches": len(vals), "min_KiB": vals[0], "max_KiB": vals[-1], "file": "profiles/" + dst}


entries = []
for steps, st_name, st_dst, tagn in ((20, "prof_driver_noextras", "r03_bench_driver_noextras_rocprofv3_kernel_stats.csv", "20"),
                                     (2000, "prof_2000", "r03_bench_2000steps_rocprofv3_kernel_stats.csv", "2000")):
    st = stats(st_name, st_dst)
    fe = counter(f"pmc_fetch_{tagn}", "FETCH_SIZE", f"r03_pmc_fetch_{tagn}_counters.csv", steps)
    wr = counter(f"pmc_write_{tagn}", "WRITE_SIZE", f"r03_pmc_write_{tagn}_counters.csv", steps)
    e = {"steps_per_launch": steps, "kernel": "rpe::" + KERNEL}
    if st:
        e.update(rocprofv3_avg_launch_us=st["avg_ns"] * 1e-3, rocprofv3_us_per_step=st["avg_ns"] * 1e-3 / steps, rocprofv3_calls=st["calls"], rocprofv3_file=st["file"])
    if fe and wr:
        traffic = 2 * fe["median_KiB"] * 1024 + wr["median_KiB"] * 1024
        e.update(traffic_bytes_per_launch=traffic, traffic_bytes_per_step=traffic / steps, FETCH_SIZE_KiB_median=fe["median_KiB"], WRITE_SIZE_KiB_median=wr["median_KiB"],
                 pmc_dispatches=[fe["dispatches"], wr["dispatches"]], pmc_files=[fe["file"], wr["file"]],
                 correction="gfx950: traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (FETCH_SIZE reads half the bytes of a 16-B/lane stream; separate --pmc passes)")
    e["source"] = "scripts/collect_evidence_r03.sh -> scripts/profile_summary_r03.py"
    e["command"] = ("rocprofv3 ... -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-hbm" if steps == 20 else
                    "rocprofv3 ... -- python3 bench.py --steps 2000 --warmup 2000 --no-extras --no-cpu-baseline --no-hbm")
    entries.append(e)
full = stats("prof_driver", "r03_bench_driver_cmd_rocprofv3_kernel_stats.csv")
index = {"normal_eq_resident_p2p_f32": entries,
         "driver_command_with_extras": None if not full else {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5",
                                                               "rocprofv3_avg_launch_us": full["avg_ns"] * 1e-3, "calls": full["calls"], "file": full["file"],
                                                               "note": "all launches of this kernel name in the run: pre-warm, calibration, timed (20 steps each), the --warmup launch (5) and the convergence leg's"}}
try:   # keep the HBM-stream entries (scripts/r03_hbm.sh) that a previous summary put into the index
    prev = json.load(open(os.path.join(prof, "r03_bench_profiles.json")))
    for k in ("hbm_stream", "hbm_stream_source"):
        if k in prev:
            index[k] = prev[k]
except Exception:
    pass
json.dump(index, open(os.path.join(prof, "r03_bench_profiles.json"), "w"), indent=1)
print(json.dumps(index, indent=1))
for name in ("bench_driver_cmd.json", "bench_default_2000steps.json", "roofline_runs.jsonl", "config4_cold_steady.jsonl", "device_loop_ab.jsonl", "pytest_gpu.txt", "pytest_new.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(prof, "r03_" + name))
