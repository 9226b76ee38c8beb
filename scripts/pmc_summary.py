"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/<tag>_pmc_traffic.json.
usage: python scripts/pmc_summary.py gpurun_out/<tag> <tag>"""
import csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def mean_counter(sub, counter, trim_to):
    f = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if "normal_eq_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    vals = [float(r["Counter_Value"]) for r in rows]
    # keep the head of the file as the committed evidence (the full trace is tens of MB)
    with open(f) as fh, open(os.path.join(root, "profiles", f"{tag}_pmc_{sub.split('_')[1]}_counter_collection.csv"), "w") as out:
        for i, line in enumerate(fh):
            if i >= trim_to: break
            out.write(line)
    vals = vals[len(vals) // 10:]  # drop the cold first launches
    return sum(vals) / len(vals), len(vals)

fetch, nf = mean_counter("pmc_fetch", "FETCH_SIZE", 60)
write, nw = mean_counter("pmc_write", "WRITE_SIZE", 60)
alg = 307200 * 26
out = {
    "normal_eq_p2p_f32_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
    "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write, "launches_averaged": [nf, nw],
    "correction": "gfx950: FETCH_SIZE reads half the bytes of a 16-B/lane coalesced stream (MI355X_MICROARCH.md, HBM); traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; the 8-B/lane mask loads are uncalibrated",
    "algorithmic_bytes_per_launch": alg,
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras",
}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out))
