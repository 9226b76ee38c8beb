#!/bin/bash
# SQ counters of the joint kernels (p2p + bearing): 10 M one launch per call, 307 200 one launch + the resident loop
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-r05_joint_sq}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in 10000000 307200; do
  RPE_PROBE_N=$n timeout 300 python3 $root/scripts/dev/joint_pmc_probe.py > $out/plain_$n.json 2>/dev/null
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    name=$(echo $set | tr ' ' '+')
    RPE_PROBE_N=$n timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p_${n}_$name -- python3 $root/scripts/dev/joint_pmc_probe.py > $out/run_${n}_$name.txt 2>&1
    f=$(ls $out/p_${n}_$name/*/*_counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && (head -1 $f; grep "joint" $f) > $out/counters_${n}_$name.csv
    rm -rf $out/p_${n}_$name
  done
done
python3 - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/counters_*.csv"):
    n = f.split("counters_")[1].split("_")[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        kind = "resident" if "resident" in k else "one_launch"
        res[n + "_" + kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
outp = {}
for k, d in res.items():
    outp[k] = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    outp[k]["dispatches"] = max(len(v) for v in d.values())
json.dump(outp, open("$out/summary.json", "w"), indent=1)
print(json.dumps(outp))
PY
cat $out/plain_*.json
